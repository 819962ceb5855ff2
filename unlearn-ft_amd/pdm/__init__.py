__version__ = "2.2.0+mi355x.r1"
