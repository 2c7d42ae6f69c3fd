"""Timing of the matrix-product shapes the feature-route gradients use (run on the GPU box: python tools/bench_matmul_shapes.py)."""
import time

import torch


def main():
    dev = torch.device("cuda:0")
    T, N, F = 512, 16384, 1568
    g = torch.randn(T, N, dtype=torch.float64, device=dev)
    ph = torch.randn(N, F, dtype=torch.float64, device=dev)
    zf = torch.randn(T, F, dtype=torch.float64, device=dev)
    def tm(name, fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter()-t0)/n*1e3:.3f} ms")
    tm("fwd zf @ ph.T", lambda: zf @ ph.T)
    tm("dph g.T @ zf", lambda: g.T @ zf)
    tm("dzf g @ ph", lambda: g @ ph)
    tm("dzf (ph.T @ g.T).T", lambda: (ph.T @ g.T).T)
    gt = g.T.contiguous()
    tm("dzf gt.T @ ph (g transposed copy)", lambda: gt.T @ ph)
    tm("g.T.contiguous()", lambda: g.T.contiguous())
    for ch in (8, 16, 32, 64):
        tm(f"dzf bmm {ch} chunks", lambda: torch.bmm(g.reshape(T, ch, N // ch).permute(1, 0, 2), ph.reshape(ch, N // ch, F)).sum(0))
    pht = ph.T.contiguous()
    tm("dzf g @ pht.T", lambda: g @ pht.T)


if __name__ == "__main__":
    main()
