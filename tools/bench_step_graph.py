#!/usr/bin/env python3
"""SVGP training step of notebooks/ts_classification.ipynb's shape (tools/bench_grad.py (a)) replayed as ONE HIP graph:
forward (HIP covariances + torch algebra + likelihood), backward (HIP gradient kernels + autograd) and the Adam update are
captured once with torch.cuda.graph and replayed per iteration; the minibatch is copied into static input tensors.
Checks that eager and replayed training produce the same ELBO trace, then times both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import kernels, models, likelihoods as LK, inducing_variables as iv

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
N, L, d, M, T, C, mb = 144, 45, 3, 4, 200, 15, 50
Xall = torch.tensor(np.cumsum(rng.standard_normal((N, L, d)) * 0.2, axis=1).reshape(N, -1), device=dev)
Yall = torch.tensor(rng.integers(0, C, (N, 1)).astype(np.float64), device=dev)
Z0 = 0.3 * rng.standard_normal((M * (M + 1) // 2, T, 2, d))


def make(trainable):
    kern = kernels.SignatureRBF(L * d, d, M, lengthscales=1.0)
    m = models.SVGPModule(kern, iv.InducingTensors(Z0.copy(), M, increments=True), LK.MultiClass(C), num_latent=C, num_data=N, device=dev)
    for p in m.kernel.parameters():
        p.requires_grad_(trainable)
    return m


def batches(n):
    g = torch.Generator(device="cpu").manual_seed(1)
    return [torch.randperm(N, generator=g)[:mb].to(dev) for _ in range(n)]


for trainable in (False, True):
    idxs = batches(60)
    # ---- eager
    m = make(trainable)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, capturable=True)
    trace_e = []
    def eager_step(idx):
        opt.zero_grad(set_to_none=True)
        loss = -m.elbo(Xall[idx], Yall[idx]); loss.backward(); opt.step()
        return loss
    for idx in idxs[:10]:
        trace_e.append(float(eager_step(idx)))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for idx in idxs[10:]:
        eager_step(idx)
    torch.cuda.synchronize(); dt_e = (time.perf_counter() - t0) / 50
    # ---- graphed
    m = make(trainable)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, capturable=True)
    xs, ys = Xall[idxs[0]].clone(), Yall[idxs[0]].clone()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    trace_g = []
    with torch.cuda.stream(side):
        for k in range(3):           # warm-up on the capture stream: scratch buffers, task lists, optimiser state
            xs.copy_(Xall[idxs[k]]); ys.copy_(Yall[idxs[k]])
            opt.zero_grad(set_to_none=True)
            loss = -m.elbo(xs, ys); loss.backward(); opt.step()
            trace_g.append(float(loss))
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g, stream=side):
        static_loss = -m.elbo(xs, ys)
        static_loss.backward()
        opt.step()
    for k in range(3, 10):
        xs.copy_(Xall[idxs[k]]); ys.copy_(Yall[idxs[k]])
        g.replay()
        trace_g.append(float(static_loss))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for idx in idxs[10:]:
        xs.copy_(Xall[idx]); ys.copy_(Yall[idx])
        g.replay()
    torch.cuda.synchronize(); dt_g = (time.perf_counter() - t0) / 50
    err = max(abs(a - b) / abs(a) for a, b in zip(trace_e, trace_g))
    print(f"SVGP training step, LIBRAS shape, minibatch {mb}, {T} inducing tensors, kernel trainable={trainable}: eager {dt_e*1e3:.2f} ms = {1/dt_e:.0f} it/s; "
          f"one HIP graph per step {dt_g*1e3:.2f} ms = {1/dt_g:.0f} it/s; ELBO traces agree to {err:.1e}")
