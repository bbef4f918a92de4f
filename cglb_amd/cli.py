"""Command line for the CGLB path on the HIP backend — mirror of the reference's cglb_experiments/cli.py for the commands
that exercise the hot path:

    python -m cglb_amd.cli -b hip -t fp64 -l LOGDIR -s SEED train -d DATASET -n STEPS cglb -k Matern32 -m cglb -i cv -M 1024
    python -m cglb_amd.cli -b hip -t fp64 -l LOGDIR metric -d DATASET cglb -k Matern32 -m cglb -i cv -M 1024 -p LOGDIR/model.json

Same option letters as cli.py:60-65, :141-152, :207-216; writes model.json / results.json / logs.json with the reference's keys
(cli.py:100-109, pytorch/interface.py:546-551).  Datasets: the reference downloads UCI sets through robustgp_experiments
(datasets.py:47-76; no network here), so DATASET is either `synthetic-N-D` (e.g. synthetic-2000-3, the generator of
cglb_amd/data.py) / `snelson-like` (N=200, D=1 stand-in for snelson1d), or a path to an .npz with arrays X, y; all are
z-normalised and split 67/33 with the seed, as datasets.py:35-39,:60-70 does.

N GPUs of one node (BASELINE config C4, "full CGLB train loop" on 8 x MI355X): launch the same command line under torch.distributed.run,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P -m cglb_amd.cli -b hip ... train ...

one process per GPU (LOCAL_RANK picks the device).  Every rank builds the same model on the same data, owns 1/N of the rows of K_ff and
of the Nystrom panel (cglb_amd/dist_context.py) and runs the same optimiser on identical (loss, gradient) pairs; rank 0 alone writes
model.json / results.json / logs.json and prints the result line.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from pathlib import Path
from typing import Callable, Tuple

import click
import numpy as np

from .backend import BACKENDS, INDUCING_VARIABLE_CONFIGS, KERNEL_CONFIGS, SGPR_CONFIGS, jsonio
from .backend.callbacks import Logger
from .data import synthetic_problem

_default_logdir = "./logs"


@dataclass
class DatasetBundle:
    name: str
    train: Tuple[np.ndarray, np.ndarray]
    test: Tuple[np.ndarray, np.ndarray]

    def to_tuple(self):
        return self.train, self.test


def _norm(x):  # datasets.py:35-39
    mu, std = x.mean(axis=0, keepdims=True), x.std(axis=0, keepdims=True)
    std = np.where(std == 0, 1.0, std)
    return (x - mu) / std


def get_dataset(name: str, seed: int = 0) -> DatasetBundle:
    if name.startswith("synthetic-"):
        _, n, d = name.split("-")
        X, y, _ = synthetic_problem(int(n), int(d), 1, seed=1234)
    elif name in ("snelson-like", "snelson1d"):
        rng = np.random.default_rng(1234)
        X = np.sort(rng.uniform(0.0, 6.0, size=(200, 1)), axis=0)
        y = (np.sin(2.0 * X[:, 0]) + 0.3 * np.cos(5.0 * X[:, 0]) + 0.15 * rng.standard_normal(200))
    elif os.path.exists(name):
        data = np.load(name)
        X, y = np.asarray(data["X"], dtype=np.float64), np.asarray(data["y"], dtype=np.float64).reshape(-1)
    else:
        raise click.BadParameter(f"unknown dataset {name!r} (no network: use synthetic-N-D, snelson-like or an .npz path)")
    X, y = _norm(X.reshape(len(X), -1)), _norm(y.reshape(-1, 1)).reshape(-1)
    perm = np.random.default_rng(seed).permutation(len(X))
    n_train = int(len(X) * 0.67)
    tr, te = perm[:n_train], perm[n_train:]
    return DatasetBundle(name, (X[tr], y[tr]), (X[te], y[te]))


def _jsonable(obj):
    if isinstance(obj, dict):
        return {k: _jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_jsonable(v) for v in obj]
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, (np.floating, np.integer)):
        return obj.item()
    return obj


def create_optimize_fn(backend, dataset: DatasetBundle, logdir: str, num_steps: int, optimizer: str) -> Callable:
    """cli.py:79-111"""

    def optimize_fn(model):
        datasets = dataset.to_tuple()
        metrics_fn = backend.metrics_fn(model, datasets)
        logger = Logger(logdir, metrics_fn, lambda: backend.model_parameters(model), 20, include_feval_log=True)
        logger.verbose = logger.verbose and _rank() == 0
        backend.optimize(model, datasets, num_steps, logger, optimizer)
        logs, results = logger.logs, metrics_fn()           # collective on N ranks: every rank evaluates the metrics
        results["id"] = logdir
        logs["id"] = logdir
        if _rank() == 0:
            backend.save(model, logdir)
            with open(Path(logdir, "results.json"), "w") as f:   # cli.py:105-109: json_tricks.dump -> same encoding (backend/jsonio.py)
                jsonio.dump(results, f)
            with open(Path(logdir, "logs.json"), "w") as f:
                jsonio.dump(logs, f)
        return results

    return optimize_fn


def create_metric_fn(backend, dataset: DatasetBundle, destination: Path) -> Callable:
    """cli.py:114-123"""

    def metric_fn(model):
        results = backend.metrics_fn(model, dataset.to_tuple())()
        results["id"] = str(destination.parent)
        if _rank() == 0:
            np.save(destination, results)
        return results

    return metric_fn


def _rank() -> int:
    return int(os.environ.get("RANK", "0"))


def _init_distributed():
    """Under torch.distributed.run (WORLD_SIZE > 1): one process per GPU, process group over RCCL ("nccl" is RCCL on ROCm).
    CGLB_DIST_BACKEND=gloo and CGLB_SHARE_GPU=1 are rehearsal knobs for a one-GPU box (several ranks on cuda:0, collectives through
    the callback provider of the library)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return
    import torch
    import torch.distributed as dist
    from datetime import timedelta
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local = 0 if os.environ.get("CGLB_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    backend = os.environ.get("CGLB_DIST_BACKEND", "nccl")
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=timedelta(minutes=5))
        else:
            dist.init_process_group(backend, timeout=timedelta(minutes=5))


@click.group()
@click.option("-b", "--backend", type=click.Choice(sorted(BACKENDS)), default="hip")
@click.option("-t", "--float-type", type=click.Choice(["fp32", "fp64"]), default="fp32")
@click.option("-l", "--logdir", type=click.Path(file_okay=False), default=_default_logdir)
@click.option("-s", "--seed", type=int, default=0)
@click.option("--keops/--no-keops", default=True)
@click.pass_context
def main(ctx, backend, float_type, logdir, seed, keops):
    logdir_path = Path(logdir).expanduser().resolve()
    logdir_path.mkdir(exist_ok=True, parents=True)
    _init_distributed()
    be = BACKENDS[backend]
    be.configure_backend(logdir=str(logdir_path), keops=keops)
    be.set_default_float(float_type)
    be.set_default_jitter(float_type)
    ctx.obj = dict(backend=be, seed=seed, logdir=str(logdir_path))


@main.group()
@click.option("-n", "--num-steps", default=100, type=int)
@click.option("-d", "--dataset", type=str, required=True)
@click.option("-o", "--optimizer", type=click.Choice(["scipy"]), default="scipy")
@click.pass_context
def train(ctx, dataset, num_steps, optimizer):
    o = ctx.obj
    bundle = get_dataset(dataset, o["seed"])
    o.update(dataset=bundle, callback=create_optimize_fn(o["backend"], bundle, o["logdir"], num_steps, optimizer))


@main.group()
@click.option("-d", "--dataset", type=str, required=True)
@click.pass_context
def metric(ctx, dataset):
    o = ctx.obj
    bundle = get_dataset(dataset, o["seed"])
    o.update(dataset=bundle, callback=create_metric_fn(o["backend"], bundle, Path(o["logdir"], "metric.npy")))


def _cglb_command(group):
    @group.command("cglb")
    @click.option("-m", "--model-class", type=click.Choice(sorted(SGPR_CONFIGS)), required=True)
    @click.option("-k", "--kernel", type=click.Choice(sorted(KERNEL_CONFIGS)), required=True)
    @click.option("-i", "--inducing-variable", type=click.Choice(sorted(INDUCING_VARIABLE_CONFIGS)), required=True)
    @click.option("-M", "--num-inducing-variables", default=100, type=int)
    @click.option("-p", "--param_file", type=click.Path(readable=True))
    @click.option("-e", "--max_error", type=float, default=1.0)
    @click.option("--vjoint/--no-vjoint", default=False)
    @click.option("--vzero/--no-vzero", default=False)
    @click.pass_context
    def cglb(ctx, model_class, kernel, inducing_variable, num_inducing_variables, param_file, max_error, vjoint, vzero):
        """cli.py:259-273 (_execute_cb_cglb)"""
        o = ctx.obj
        cfg = SGPR_CONFIGS[model_class](KERNEL_CONFIGS[kernel](), INDUCING_VARIABLE_CONFIGS[inducing_variable](num_inducing_variables),
                                        max_error, vjoint, vzero)
        model = o["backend"].create_model(cfg, o["dataset"].train)
        if param_file:
            model = o["backend"].load(model, param_file)
        results = o["callback"](model)
        if _rank() == 0:
            click.echo(json.dumps(_jsonable(results)))

    return cglb


_cglb_command(train)
_cglb_command(metric)

if __name__ == "__main__":
    main()
