"""Where does the bf16 engine's distance from the fp32 oracle come from?  

At FIXED weights (no optimiser in the loop, so nothing chaotic): relative L2 error against the fp32 CPU oracle of
  * the nine hooked block activations and the prediction of the student forward,
  * the teacher's prediction,
  * the four loss heads of one main step and one upper step,
  * the whole gradient vector of the main step (per tensor: the worst and the norm-weighted mean),
for (a) the HIP bf16 engine, (b) the oracle under the CPU autocast (mixed=True), (c) the oracle under CUDA autocast's op policy
(mixed="cuda": fp32 norms, fused-SDPA numerics - what the reference trains under).  Then the 12-point loss curve of
tests/test_step_parity_gpu.py for all three.  Prints one table; run on the GPU box:  python tools/bf16_bisect.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "unlearn-ft_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from pdm_ref import step as ostep, unet as ounet, weights as oweights   # noqa: E402
import test_step_parity_gpu as T   # noqa: E402


def l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    from pdm.training.bilevel import BilevelStepper
    ocfg, dense, psd, info, student, teacher = T._setup(torch.bfloat16, drop_depth=(1, 9))
    lat, noise, t, ehs, empty = T._inputs()
    ac = ostep.alphas_cumprod()
    tinfo = oweights.dense_info(ocfg)
    noisy = ostep.add_noise(ac, lat, noise, t)

    # ---- forward at fixed weights
    res = {}
    for name, mixed in (("fp32", False), ("cpu-autocast", True), ("cuda-policy", "cuda")):
        mp = ostep._mixed(mixed)
        acts = {}
        with torch.no_grad(), mp.ctx():
            pred = ounet.unet_forward(psd, ocfg, info, noisy, t, ehs, acts)
            tp = ounet.unet_forward(mp.teacher_sd(dense), ocfg, tinfo, noisy, t, ehs)
        res[name] = dict({k: v.float() for k, v in acts.items()}, pred=pred.float(), teacher=tp.float())
    acts = {}
    for i, h in enumerate(student.down_blocks):
        h.register_forward_hook(lambda m, inp, out, i=i: acts.__setitem__(f"d{i}", out[0]))
    student.mid_block.register_forward_hook(lambda m, inp, out: acts.__setitem__("m", out))
    for i, h in enumerate(student.up_blocks):
        h.register_forward_hook(lambda m, inp, out, i=i: acts.__setitem__(f"u{i}", out))
    out = student.eval()(noisy, t, ehs).sample.float().cpu()
    hip = dict({k: v.float().cpu() for k, v in acts.items()}, pred=out, teacher=teacher(noisy, t, ehs).sample.float().cpu())
    student.train()
    print("== forward at fixed weights: relative L2 error vs the fp32 oracle")
    print(f"{'tensor':10s} {'HIP bf16':>10s} {'cpu-autocast':>13s} {'cuda-policy':>12s}   HIP / cuda-policy")
    for key in list(ostep.BLOCK_KEYS) + ["pred", "teacher"]:
        r = res["fp32"][key]
        e = [l2(hip[key], r), l2(res["cpu-autocast"][key], r), l2(res["cuda-policy"][key], r)]
        print(f"{key:10s} {e[0]:10.3e} {e[1]:13.3e} {e[2]:12.3e}   {e[0] / e[2]:.2f}")

    # ---- one main step: loss heads and gradients
    grads, heads = {}, {}
    for name, mixed in (("fp32", False), ("cpu-autocast", True), ("cuda-policy", "cuda")):
        P = {k_: v.clone().requires_grad_(True) for k_, v in psd.items()}
        o = ostep.main_step_loss((P, info), (dense, tinfo), ocfg, ac, lat, noise, t, ehs, mixed=mixed)
        o[0].float().backward()
        heads[name] = [float(x.detach()) for x in o[:4]]
        grads[name] = {k_: v.grad.float() for k_, v in P.items()}
        ou = ostep.upper_step_loss((psd, info), (dense, tinfo), ocfg, ac, lat, noise, t, ehs, empty, mixed=mixed)
        heads[name].append(float(ou[0].detach()))
    st = BilevelStepper(student, teacher, lr=2e-5, upper_lr=5e-5, bilevel=True)
    L = st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda())
    torch.cuda.synchronize()
    heads["hip"] = list(st.total(L))
    grads["hip"] = {k_: v.float().cpu() for k_, v in student.store.state_dict(arena=student.store.grad).items()}
    from pdm import _pdmk as k
    k.zero_(student.store.grad)
    Lu = st.upper_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda(), empty.cuda(), backward=False)
    heads["hip"].append(st.total(Lu, upper=True)[0])
    print("== loss heads (total, diff, dist, block, upper total): relative error vs fp32")
    for name in ("hip", "cpu-autocast", "cuda-policy"):
        print(f"{name:13s}", " ".join(f"{abs(a - b) / max(abs(b), 1e-9):9.2e}" for a, b in zip(heads[name], heads["fp32"])))
    print("== main-step gradients: relative L2 error vs fp32 (whole vector | worst tensor of >= 1024 elements)")
    for name in ("hip", "cpu-autocast", "cuda-policy"):
        ga = torch.cat([grads[name][k_].flatten() for k_ in sorted(grads["fp32"])])
        gr = torch.cat([grads["fp32"][k_].flatten() for k_ in sorted(grads["fp32"])])
        worst = max((l2(grads[name][k_], grads["fp32"][k_]), k_) for k_ in grads["fp32"] if grads["fp32"][k_].numel() >= 1024)
        print(f"{name:13s} {l2(ga, gr):9.3e} | {worst[0]:9.3e} {worst[1]}")
    # by layer kind
    kinds = ("conv1.weight", "conv2.weight", "to_q.weight", "to_k.weight", "to_v.weight", "to_out.0.weight", "ff.net.0.proj.weight",
             "ff.net.2.weight", "proj_in.weight", "proj_out.weight", "norm1.weight", "norm2.weight", "norm3.weight", "norm.weight",
             "time_emb_proj.weight", "conv_shortcut.weight", "bias")
    print(f"{'kind':24s} {'HIP':>10s} {'cpu-ac':>10s} {'cuda-pol':>10s}")
    for kd in kinds:
        names = [n for n in grads["fp32"] if n.endswith(kd)]
        if not names:
            continue
        cat = lambda w: torch.cat([grads[w][n].flatten() for n in names])
        r = cat("fp32")
        print(f"{kd:24s} {l2(cat('hip'), r):10.3e} {l2(cat('cpu-autocast'), r):10.3e} {l2(cat('cuda-policy'), r):10.3e}")

    # ---- the loss curve of the test (optimiser in the loop)
    del st
    ocfg, dense, psd, info, student, teacher = T._setup(torch.bfloat16, drop_depth=(1, 9))
    lr, ulr = 2e-5, 5e-5
    T._CURVE.clear()
    ref = T._oracle_curve(ocfg, dense, psd, info, lr, ulr)
    mixc = T._oracle_curve(ocfg, dense, psd, info, lr, ulr, mixed=True)
    mixu = T._oracle_curve(ocfg, dense, psd, info, lr, ulr, mixed="cuda")
    st = BilevelStepper(student, teacher, lr=lr, upper_lr=ulr, bilevel=True)
    hipc = []
    for it, (lat, noise, t, ehs, empty) in enumerate(T._curve_inputs()):
        for name in ["main"] + (["upper"] if (it + 1) % 3 == 0 else []):
            if name == "main":
                L = st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda())
            else:
                L = st.upper_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda(), empty.cuda())
            st.optimizer_step(upper=name == "upper")
            hipc.append(st.total(L, upper=name == "upper")[0])
    rel = lambda a: [abs(x - r) / max(abs(r), 1e-6) for x, r in zip(a, ref)]
    rms = lambda d: (sum(x * x for x in d) / len(d)) ** 0.5
    print("== 12-point loss curve, relative distance from the fp32 oracle (max | rms | per point)")
    for name, c in (("hip", hipc), ("cpu-autocast", mixc), ("cuda-policy", mixu)):
        d = rel(c)
        print(f"{name:13s} {max(d):.2e} | {rms(d):.2e} |", " ".join(f"{x:.1e}" for x in d))
    print("raw fp32 curve:", " ".join(f"{x:.5f}" for x in ref))


if __name__ == "__main__":
    main()
