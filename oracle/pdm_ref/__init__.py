"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the bilevel fine-tune/unlearn step of rezashkv/unlearn-ft.

Pure-torch fp32 restatement of the reference hot path. Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; the product
(``unlearn-ft_amd/``) never does and fails loudly when its HIP library is missing.

Parity status: the reference has no tests/golden vectors for this path and its own
model/trainer modules cannot be imported here (``diffusers``/``torchvision`` absent - ordinary
ModuleNotFoundError, nothing was denied).  The restatement is therefore pinned sub-block by sub-block
against the reference pieces that DO import (``oracle/validate_against_reference.py``): compute_snr,
hard_concrete, gates, and the vendored CompVis ResBlock / SpatialTransformer / Downsample / Upsample /
timestep_embedding / make_beta_schedule twins; known-answer vectors are committed under tests/golden/.
``vae.py`` (the VAE encode in front of the step) is pinned the same way against the vendored CompVis
Encoder / AttnBlock / Downsample / DiagonalGaussianDistribution (``oracle/validate_vae_against_reference.py``);
``clip_text.py`` (row N2, the text encoder of the dataset transform) against ``transformers.CLIPTextModel`` itself, the
third-party class the reference instantiates (``oracle/validate_clip_against_transformers.py``).
``sampler.py`` (row N3): the VAE decoder is pinned like the encoder; the PNDM scheduler is PARITY UNPINNED (diffusers
absent, no vendored twin) - restated from the published algorithm and checked against the closed forms it must reproduce.
"""
from .config import UNetConfig  # noqa: F401
