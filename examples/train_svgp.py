"""Train a sparse variational GP classifier with a signature kernel on the GPU (synthetic data).

The reference's counterpart is notebooks/ts_classification.ipynb (GPflow SVGP + SignatureRBF + InducingTensors, trained with
TensorFlow optimisers); here the same model is `gpsig_amd.models.SVGPModule`, whose kernel evaluations and their gradients
run in the HIP library.

    python examples/train_svgp.py [--iterations 200] [--kernel rbf|linear] [--order 1] [--low-rank] [--graph]

--kernel linear --order 4 is the signature kernel proper (the variant the reference's notebook validates against esig); --low-rank
trains through the Nystrom / sparse-projection features (low_rank=True of the reference), whose reverse pass is a HIP kernel too.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from gpsig_amd import inducing_variables, kernels, likelihoods, models, utils


def make_data(rng, n, length, classes):
    """Noisy random walks whose first coordinate carries a class-dependent oscillation; a time coordinate is appended."""
    t = np.linspace(0.0, 1.0, length)
    y = rng.integers(0, classes, n)
    X = np.cumsum(0.1 * rng.standard_normal((n, length, 2)), axis=1)
    X[:, :, 0] += np.sin(2 * np.pi * (1 + y)[:, None] * t[None, :])
    X = np.concatenate([X, np.broadcast_to(t[None, :, None], (n, length, 1))], axis=2)
    return X, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=200)
    ap.add_argument("--num-inducing", type=int, default=64)
    ap.add_argument("--minibatch", type=int, default=64)
    ap.add_argument("--graph", action="store_true", help="record the training step as one HIP graph (SVGPModule.fit(graph=True))")
    ap.add_argument("--kernel", choices=["rbf", "linear"], default="rbf")
    ap.add_argument("--order", type=int, default=1, help="1: the reference's default; num_levels: the signature of the piecewise-linear path")
    ap.add_argument("--low-rank", action="store_true", help="low_rank=True (Nystrom features + sparse projections, default ranks 20)")
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    classes, length, levels = 3, 40, 4
    Xtr, ytr = make_data(rng, 300, length, classes)
    Xte, yte = make_data(rng, 200, length, classes)
    d = Xtr.shape[2]

    cls = kernels.SignatureRBF if args.kernel == "rbf" else kernels.SignatureLinear
    extra = dict(low_rank=True, num_components=20, rank_bound=20) if args.low_rank else {}
    kern = cls(length * d, d, levels, lengthscales=utils.suggest_initial_lengthscales(Xtr, 1000, rng=rng), order=args.order, **extra)
    kern.rng = np.random.default_rng(1)
    Z = utils.suggest_initial_inducing_tensors(Xtr, levels, args.num_inducing, labels=ytr, increments=True, rng=rng)
    feat = inducing_variables.InducingTensors(Z, levels, increments=True)
    model = models.SVGPModule(kern, feat, likelihoods.MultiClass(classes), num_latent=classes, num_data=Xtr.shape[0], device="cuda:0")

    dev = torch.device("cuda:0")
    X = torch.tensor(Xtr.reshape(len(Xtr), -1), device=dev)
    Y = torch.tensor(ytr[:, None].astype(np.float64), device=dev)
    trace = model.fit(X, Y, iterations=args.iterations, lr=2e-2, minibatch_size=args.minibatch,
                      callback=lambda it, elbo: print(f"iteration {it:4d}  ELBO {elbo:10.2f}") if it % 25 == 0 else None,
                      graph=args.graph)
    with torch.no_grad():
        p, _ = model.predict_y(torch.tensor(Xte.reshape(len(Xte), -1), device=dev))
    acc = float((p.argmax(dim=1).cpu().numpy() == yte).mean())
    print(f"final ELBO {trace[-1]:.2f}; test accuracy {acc:.3f}")
    model.kernel.write_back()       # the trained hyper-parameters now also drive kern.K(...) etc.
    return acc


if __name__ == "__main__":
    main()
