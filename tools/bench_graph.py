"""Latency of small evaluations: eager C-ABI calls against a recorded HIP graph (gpsig_graph_begin / _launch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import kernels as K

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)


def timeit(fn, n=300, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


for (name, N, L, d, M, T) in (("BASELINE configs[0] shape", 64, 32, 3, 4, 0), ("SVGP minibatch (LIBRAS shape)", 50, 45, 3, 4, 200), ("minibatch 256, L=64, d=8, M=5", 256, 64, 8, 5, 512)):
    kern = K.SignatureRBF(L * d, d, M, lengthscales=1.0)
    X = torch.tensor(rng.standard_normal((N, L * d)) * 0.3, device=dev)
    cases = [("K(X)", "K", (X,)), ("Kdiag(X)", "Kdiag", (X,))]
    if T:
        Z = torch.tensor(rng.standard_normal((M * (M + 1) // 2, T, d)) * 0.3, device=dev)
        cases += [("K_tens(Z)", "K_tens", (Z,)), ("K_tens_vs_seq(Z, X)", "K_tens_vs_seq", (Z, X))]
    for label, m, a in cases:
        g = kern.graphed(m, *a)
        te = timeit(lambda: getattr(kern, m)(*a))
        tg = timeit(g.replay)
        tl = timeit(g.graph.launch)
        print(f"{name}: {label}: eager {te:.1f} us, graph replay {tg:.1f} us (launch only {tl:.1f} us)")
