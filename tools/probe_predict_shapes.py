"""Prediction at the reference's shapes: models.SVGPModule.predict_f over a test set of N sequences (default 2,000) at each data set's columns / length
(tools/reference_shapes.py), in chunks as a user would run it.  Times per 1,000 sequences.  python tools/probe_predict_shapes.py [datasets] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import reference_shapes as RS
names = (sys.argv[1] if len(sys.argv) > 1 else "ECG,UWave,NetFlow,Wafer,ArabicDigits,AUSLAN,CMUsubject16").split(",")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device("cuda:0")
for name in names:
    s, model, X, Y = RS.build(name, dev)
    rng = np.random.default_rng(1)
    d, L = s["d"], s["L"]
    Xt = np.cumsum(rng.standard_normal((N, L, d)) / np.sqrt(L), axis=1)
    Xt[:, :, 0] = np.linspace(0.0, 1.0, L)[None, :]
    Xt = torch.as_tensor(Xt.reshape(N, -1), device=dev)
    def run():
        with torch.no_grad():
            return model.predict_f(Xt)
    try:
        run(); torch.cuda.synchronize(); t0 = time.perf_counter(); mu, var = run(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        print("%-22s %4d columns L = %3d: predict_f over %d sequences %8.2f ms (%.2f ms per 1,000), finite %s"
              % (name, s["d_eff"], L, N, dt, dt * 1000 / N, bool(torch.isfinite(mu).all() and torch.isfinite(var).all())), flush=True)
    except Exception as e:
        print("%-22s FAILED %s: %s" % (name, type(e).__name__, str(e)[:200]), flush=True)
