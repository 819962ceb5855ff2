"""CPU: host-side logic of the product (no kernel launches): shape table vs the oracle's literal pruning, MAC counter,
packed-arena import/export round trip, LR schedule, C-ABI surface."""
import os
import re

import torch

from pdm_ref import arch as oarch, weights as oweights, unet as ounet
from pdm_ref.config import UNetConfig as OCfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_exports_every_declared_symbol():
    from pdm import _pdmk
    hdr = open(os.path.join(ROOT, "include", "pdmk.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t) (pdmk_\w+)\(", hdr, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(_pdmk.EXPORTS), declared ^ set(_pdmk.EXPORTS)
    for name in declared:
        assert hasattr(_pdmk._lib, name)
    assert _pdmk.version() >= 100


def test_workspace_queries_and_plan_file_roundtrip(tmp_path):
    """The *_workspace_bytes() queries answer without a GPU; the plan cache round-trips through its file."""
    from pdm import _pdmk as k
    L = k._lib
    assert L.pdmk_gemm_splitk_workspace_bytes(2048, 1280, 3) == 3 * 2048 * 1280 * 4
    assert L.pdmk_gemm_splitk_workspace_bytes(0, 8, 1) == -1
    assert L.pdmk_groupnorm_workspace_bytes(8, 32) == 8 * 32 * 64 * 8
    assert L.pdmk_groupnorm_bwd_part_workspace_bytes(32, 80) == 2048 * 2 * 2560 * 4
    assert L.pdmk_layernorm_bwd_part_workspace_bytes(32768, 320) == (32768 // 16 + 1) * 2 * 320 * 4
    assert L.pdmk_attn_bwd_workspace_bytes(8, 5, 4096, 77) == 2 * 16 * 8 * 5 * 77 * 64 * 4
    assert L.pdmk_attn_bwd_workspace_bytes(8, 5, 4096, 4096) == 0
    f = tmp_path / "plans.txt"
    f.write_text(f"pdmk-plan {k.version()}\nc 1 2 3 0 0 0 0 0 0 1 7\ns 1 2 3 0 0 0 0 0 0 0 2\n")
    k.plan_clear()
    assert k.plan_import(str(f)) == 2 and k.plan_size() == 2
    g = tmp_path / "out.txt"
    k.plan_export(str(g))
    assert sorted(g.read_text().splitlines()) == sorted(f.read_text().splitlines())
    (tmp_path / "old.txt").write_text("pdmk-plan 1\nc 1 2 3 0 0 0 0 0 0 1 7\n")
    try:
        k.plan_import(str(tmp_path / "old.txt"))
        raise AssertionError("a plan file of another build must be refused")
    except k.PdmkError:
        pass
    k.plan_clear()
    assert k.plan_size() == 0


def test_partial_dims_queries_and_deferred_queues_chunk_by_32(monkeypatch):
    """The geometry queries of the deferred reductions answer without a GPU and agree with the workspace queries; the two
    queues hand the library at most 32 items per launch, keep their slabs alive until then, and the slab queue reports
    itself full by count or by bytes (host logic only: the entry points are replaced by recorders)."""
    import ctypes as C
    from pdm import _pdmk as k
    nblk, n = k._dims(k._lib.pdmk_layernorm_bwd_partial_dims, 32768, 320)
    assert (nblk, n) == (512, 320) and nblk * 2 * n * 4 <= k._lib.pdmk_layernorm_bwd_part_workspace_bytes(32768, 320)   # ~512 blocks of >= 32 rows
    nblk, n = k._dims(k._lib.pdmk_groupnorm_bwd_partial_dims, 8, 4096, 320, 32, 10, k.BF16)
    assert n == 320 and 0 < nblk * 2 * n * 4 <= k._lib.pdmk_groupnorm_bwd_part_workspace_bytes(32, 10)
    a, b = C.c_int32(), C.c_int32()
    assert k._lib.pdmk_layernorm_bwd_partial_dims(0, 320, C.byref(a), C.byref(b)) == -1
    assert k._lib.pdmk_groupnorm_bwd_partial_dims(8, 64, 321, 32, 10, k.BF16, C.byref(a), C.byref(b)) == -1     # C % 8
    calls = []
    monkeypatch.setattr(k, "_st", lambda: None)
    monkeypatch.setattr(k._lib, "pdmk_reduce_partials_group", lambda arr, cnt, st: calls.append(("p", cnt)) or 0)
    monkeypatch.setattr(k._lib, "pdmk_splitk_finish_group", lambda arr, cnt, st: calls.append(("s", cnt)) or 0)
    g0, g1 = torch.zeros(8), torch.zeros(8)
    q = k.PartialQueue()
    slabs = [q.slab(torch.device("cpu"), 4, 8, g0, g1) for _ in range(70)]
    assert calls == [("p", 32), ("p", 32)] and len(q.items) == 6 and all(s.numel() == 4 * 2 * 8 for s in slabs)
    q.flush()
    assert calls[-1] == ("p", 6) and not q.items
    calls.clear()
    sq = k.SlabQueue(max_bytes=1000)
    assert not sq.full()
    sq.add(torch.zeros(200), g0, 8, 25)
    assert not sq.full()
    sq.add(torch.zeros(100), g0, 8, 12)
    assert sq.full()                       # 1200 bytes queued
    for _ in range(40):
        sq.add(torch.zeros(4), g0, 4, 1)
    sq.flush()
    assert calls == [("s", 32), ("s", 10)] and sq.bytes == 0 and not sq.items


def test_structure_matches_oracle_and_reference_sizes():
    from pdm.models.unet import spec
    for mk, omk in ((spec.UNetConfig.tiny, OCfg.tiny), (spec.UNetConfig.sd21, OCfg.sd21)):
        assert spec.gate_structure(mk()) == oarch.structure(omk())
    assert spec.arch_vector_size(spec.UNetConfig.sd21()) == 1620
    g = torch.Generator().manual_seed(0)
    av = spec.get_random_arch_vector(0.5, spec.gate_structure(spec.UNetConfig.sd21()), g)
    assert av.shape == (1, 1620) and all(min(abs(v), abs(v - 0.9)) < 1e-6 for v in av.unique().tolist())


def test_plan_macs_equals_oracle_mac_counter():
    from pdm.models.unet import spec
    cfg, ocfg = spec.UNetConfig.tiny(), OCfg.tiny()
    dense = oweights.init_dense_state_dict(ocfg, 0)
    av = oarch.random_arch_vector(ocfg, 0.55, 0, drop_depth=(1, 5, 9, 12))
    psd, info = oweights.prune_state_dict(dense, ocfg, av)
    assert spec.plan_macs(cfg, spec.apply_arch_vector(cfg, av), 16, 13)[0] == ounet.unet_macs(psd, ocfg, info, 16, 13)
    assert spec.plan_macs(cfg, spec.build_blocks(cfg), 16, 13)[0] == ounet.unet_macs(dense, ocfg, oweights.dense_info(ocfg), 16, 13)
    total, by = spec.plan_macs(spec.UNetConfig.sd21(), spec.build_blocks(spec.UNetConfig.sd21()), 64, 77)
    assert abs(total / 1e9 - 402.13) < 0.01 and abs(by["conv3x3"] / total - 0.498) < 0.002
    av55, ratio, _ = spec.arch_vector_for_budget(spec.UNetConfig.sd21(), 0.55)
    assert abs(ratio - 0.55) < 0.01


def test_packed_arena_roundtrip_is_the_reference_pruning():
    from pdm.models.unet import spec, params
    from pdm.models.unet.unet_2d_conditional import slice_dense_state_dict
    cfg, ocfg = spec.UNetConfig.tiny(), OCfg.tiny()
    dense = oweights.init_dense_state_dict(ocfg, 0)
    av = oarch.random_arch_vector(ocfg, 0.55, 0, drop_depth=(1, 5, 9, 12))
    psd, _ = oweights.prune_state_dict(dense, ocfg, av)
    blocks = spec.apply_arch_vector(cfg, av)
    mine = slice_dense_state_dict(dense, cfg, blocks)
    assert set(mine) == set(psd)
    assert all(torch.equal(mine[k], psd[k]) for k in psd)
    store = params.ParamStore(params.build_entries(cfg, blocks), "cpu", torch.float32, train=False)
    store.load_state_dict(psd, refresh=False)
    back = store.state_dict()
    assert set(back) == set(psd) and all(torch.equal(back[k], psd[k]) for k in psd)
    # every packed dim is a multiple of the channel pad and the padding is zero
    for e in store.entries:
        assert all(d % spec.CHANNEL_PAD == 0 for d in e.shape if d != 9), (e.key, e.shape)
    logical = sum(v.numel() for v in psd.values())
    assert store.num_logical_params() == logical
    tot = float(sum(v.double().abs().sum() for v in psd.values()))
    assert abs(float(store.master.double().abs().sum()) - tot) <= 1e-9 * tot        # padding is exactly zero


def test_lr_schedule_constant_with_warmup_times_world():
    from pdm.training.bilevel import FusedAdamW

    class S:            # stand-in store: FusedAdamW only needs .master for sizing here
        master = torch.zeros(8)
    opt = FusedAdamW(S, lr=1e-3, warmup_steps=250 * 4, sched_mult=4)     # trainer.py:436-443 with W=4
    lrs = []
    for _ in range(300):
        lrs.append(opt.current_lr())
        opt.sched_k += opt.sched_mult
    assert lrs[0] == 0.0 and abs(lrs[125] - 0.5e-3) < 1e-9 and lrs[250] == 1e-3 and lrs[-1] == 1e-3


def test_cli_flags_of_reference_slurm_scripts_parse():
    from pdm.utils.arg_utils import parse_args
    a = parse_args(["--base_config_path", "x.yaml", "--cache_dir", "/c", "--wandb_run_name", "r", "--pruning_ckpt_dir",
                    "/p", "--expert_id", "5"])
    assert a.expert_id == 5 and a.seed == 43 and a.pretrained_model_name_or_path == "stabilityai/stable-diffusion-2-1"


def test_committed_bench_line_has_the_contract_fields():
    """profiles/r01_bench.json is a verbatim bench.py line: every field of the measurement contract must be present."""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "images/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "two_sided"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port")
    assert abs(d["value"] - d["steps"] * d["config"]["global_batch"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 0.05 * d["value"]


def test_empty_batch_rule_follows_the_reference():
    """trainer.py:2771-2772 skips a batch when `pixel_values.numel() == 0` (collate_fn's all-samples-failed case)."""
    import torch
    from pdm.training.trainer import Trainer
    z = torch.zeros(0)
    assert Trainer._is_empty({"pixel_values": z, "prompt_embeds": torch.zeros(2, 77, 1024)})
    assert not Trainer._is_empty({"pixel_values": torch.zeros(1, 3, 8, 8), "prompt_embeds": z})
    assert Trainer._is_empty({"latents": z}) and not Trainer._is_empty({"latents": torch.zeros(1, 4, 8, 8)})
    assert Trainer._is_empty({"input_ids": torch.zeros(0, 77, dtype=torch.int64)}) and Trainer._is_empty({})


def test_optimizer_state_uses_the_torch_adamw_layout_in_reference_parameter_order():
    """optimizer.bin / scheduler.bin interchange (trainer.py:452-514): FusedAdamW.state_dict() is what
    torch.optim.AdamW.state_dict() would be for the reference module - loadable by a torch AdamW over parameters in the
    reference's registration order - and it round-trips through the packed arenas bit-exactly."""
    from pdm.models.unet import spec
    from pdm.models.unet.params import ParamStore, build_entries, reference_param_order
    from pdm.training.bilevel import FusedAdamW
    cfg = spec.UNetConfig.tiny()
    av = oarch.random_arch_vector(OCfg.tiny(), 0.6, seed=3, drop_depth=(1, 9))
    blocks = spec.apply_arch_vector(cfg, av)
    store = ParamStore(build_entries(cfg, blocks), "cpu", torch.float32, train=True)
    dense = oweights.init_dense_state_dict(OCfg.tiny(), seed=0)
    psd, _ = oweights.prune_state_dict(dense, OCfg.tiny(), av)
    store.load_state_dict(psd, refresh=False)
    names = store.state_dict_names()
    order = reference_param_order(names)
    assert sorted(order) == sorted(psd) and order[0] == "conv_in.weight" and order[-1] == "conv_out.bias"
    # attentions before resnets inside a block, up_blocks before mid_block (registration order of the reference module)
    assert order.index("down_blocks.0.attentions.0.norm.weight") < order.index("down_blocks.0.resnets.0.norm1.weight")
    assert order.index("up_blocks.3.resnets.0.norm1.weight") < order.index("mid_block.attentions.0.norm.weight")
    opt = FusedAdamW(store, 3e-4, warmup_steps=10)
    g = torch.Generator().manual_seed(1)
    # random moments inside the logical (unpadded) region only: export + import of a state dict cannot carry padding
    m_sd = {n: torch.randn(v.shape, generator=g) for n, v in psd.items()}
    v_sd = {n: torch.rand(v.shape, generator=g) for n, v in psd.items()}
    store.load_state_dict(m_sd, arena=opt.m)
    store.load_state_dict(v_sd, arena=opt.v)
    opt.t, opt.sched_k = 3, 3
    sd = opt.state_dict()
    params = [torch.nn.Parameter(psd[n].clone()) for n in order]
    topt = torch.optim.AdamW(params, lr=3e-4, weight_decay=0.0)
    topt.load_state_dict(sd)                                     # torch accepts it as its own
    assert float(topt.state[params[5]]["step"]) == 3.0
    assert torch.equal(topt.state[params[5]]["exp_avg"], m_sd[order[5]])
    assert abs(topt.param_groups[0]["lr"] - 3e-4 * 0.3) < 1e-12 and topt.param_groups[0]["initial_lr"] == 3e-4
    sch = torch.optim.lr_scheduler.LambdaLR(topt, lambda k_: min(1.0, k_ / 10.0))
    sch.load_state_dict(opt.scheduler_state_dict())
    assert sch.last_epoch == 3
    # and what torch writes comes back into the packed arenas
    opt2 = FusedAdamW(store, 3e-4, warmup_steps=10)
    opt2.load_state_dict(topt.state_dict())
    opt2.load_scheduler_state_dict(sch.state_dict())
    assert opt2.t == 3 and opt2.sched_k == 3 and torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)
    fresh = FusedAdamW(store, 3e-4)
    assert fresh.state_dict()["state"] == {}                     # no step taken: empty state, like torch
    opt2.load_state_dict(fresh.state_dict())
    assert opt2.t == 0 and float(opt2.m.abs().max()) == 0.0
