#!/usr/bin/env python3
"""One SVGP training step (-ELBO forward, backward, Adam update) at the reference's own run settings (tools/reference_shapes.py) replayed as ONE HIP graph
against the eager step: what the host side (Python, torch dispatch, ~200 short kernels) costs per step, data set by data set.
    python tools/bench_step_graph_shapes.py [all|<dataset>,..]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import reference_shapes as RS  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "all"
names = sorted(RS.DATASETS, key=lambda n: (RS.shape_of(n)["d_eff"], RS.shape_of(n)["L"])) if which == "all" else which.split(",")
for name in names:
    try:
        res = {}
        for mode in ("eager", "graph"):
            s, m, X, Y = RS.build(name, dev)
            opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, capturable=True)
            xs, ys = X.clone(), Y.clone()

            def step():
                opt.zero_grad(set_to_none=True)
                loss = -m.elbo(xs, ys)
                loss.backward()
                opt.step()
                return loss
            trace = []
            if mode == "eager":
                for _ in range(4):
                    trace.append(float(step()))
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20):
                    step()
                torch.cuda.synchronize()
                res[mode] = ((time.perf_counter() - t0) / 20 * 1e3, trace)
            else:
                side = torch.cuda.Stream(dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    for _ in range(3):                      # warm-up on the capture stream: scratch buffers, rocBLAS workspace, optimiser state
                        trace.append(float(step()))
                torch.cuda.current_stream(dev).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                opt.zero_grad(set_to_none=True)
                with torch.cuda.graph(g, stream=side):
                    static_loss = -m.elbo(xs, ys)
                    static_loss.backward()
                    opt.step()
                g.replay()
                trace.append(float(static_loss))
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20):
                    g.replay()
                torch.cuda.synchronize()
                res[mode] = ((time.perf_counter() - t0) / 20 * 1e3, trace)
        err = max(abs(a - b) / abs(a) for a, b in zip(res["eager"][1], res["graph"][1]))
        print("%-22s %5d columns, L = %3d: eager step %7.2f ms, one HIP graph per step %7.2f ms; -ELBO traces agree to %.1e"
              % (name, s["d_eff"], s["L"], res["eager"][0], res["graph"][0], err), flush=True)
    except Exception as e:      # noqa: BLE001
        print("%-22s failed: %s: %s" % (name, type(e).__name__, str(e)[:300]), flush=True)
