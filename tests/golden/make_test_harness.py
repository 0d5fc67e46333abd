"""Generates tests/golden/test_harness.npz by calling the REFERENCE DRIVER's own ``test()`` function
(train_vidreid_xent_htri.py:450-542 of weleen/AGRL.pytorch, mounted read-only at /root/reference) on the reference's own
``vmgn`` model, on the CPU, over small synthetic loaders. Build container only:  python tests/golden/make_test_harness.py

SURVEY 8(a) row 13: features, distance matrix and ranking each have reference-generated fixtures (make_golden.py); this one pins
the whole dataflow of ``test()`` -- eval forward per batch, (dense samplers) clip pooling, concatenation, compute_distance_matrix,
evaluate_rank(use_metric_mars=True) -- END TO END on what the reference itself returns.

The driver is loaded by file path (nothing of it is copied and it never travels). Its module level imports its whole control plane
and parses the command line, so before loading it this harness
  * sets ``sys.argv`` to the options of the case,
  * installs stub modules for what the container lacks or what is out of scope: ``tensorboardX``, ``h5py``, ``torchvision``,
    ``torchreid.transforms``, ``torchreid.data_manager`` / ``lr_scheduler`` / ``optimizers`` (never called by ``test()``), and a
    ``torchreid.models`` whose ``init_model`` is the reference's own ``vmgn`` factory loaded by path (the reference's
    ``models/__init__.py`` imports sibling models that need torchvision),
  * installs the ``sklearn.metrics.base`` shim of make_golden.py so that the reference's ``torchreid.metrics`` package imports.
``torchreid.metrics`` (distance.py, rank.py), ``torchreid.utils.*`` and ``torchreid.samplers`` are the reference's own files.
Inputs are rebuilt from seeds on the test side (tests/recipe.py, tests/harness_split.py); only the split's labels, the BNNeck
calibration and the expected outputs are written."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("AGRL_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)  # the reference's torchreid package; this build's package is NOT on the path

from recipe import calibrate_bnneck, recipe_state_dict  # noqa: E402
import harness_split as HS  # noqa: E402
import make_golden as MG  # noqa: E402


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_driver(argv):
    MG.install_shims()
    ref_vmgn = MG.load("ref_vmgn", "torchreid/models/vmgn.py")
    ref_vmgn.init_pretrained_weights = lambda *a, **k: None
    stub("tensorboardX", SummaryWriter=object)
    stub("h5py")
    stub("torchreid.data_manager", get_names=lambda: ["mars"], init_dataset=None)
    stub("torchreid.lr_scheduler")
    stub("torchreid.optimizers", init_optim=None)
    stub("torchreid.models", get_names=lambda: ["vmgn"], init_model=lambda name, *a, **k: ref_vmgn.vmgn(*a, **k))
    old = sys.argv
    sys.argv = ["train_vidreid_xent_htri.py"] + argv
    try:
        drv = MG.load("ref_driver", "train_vidreid_xent_htri.py")
    finally:
        sys.argv = old
    return drv, ref_vmgn


def main():
    torch.set_num_threads(8)
    drv, ref_vmgn = load_driver(["--use-cpu", "--test-sample", "evenly", "--seq-len", str(HS.S), "--test-batch", "8",
                                 "--dist-metric", "cosine", "-a", "vmgn"])
    q_pids, q_cams, g_pids, g_cams = HS.make_split()
    model = ref_vmgn.vmgn(num_classes=HS.N_ID, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                          pyramid_part=True, use_pose=True, learn_graph=True, consistent_loss=False)
    sd = recipe_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd)
    model.eval()
    # BNNeck calibration (as training would leave it): from the reference model's own pre-BN features of the split
    feats = {"g": [], "a": []}
    h1 = model.global_bottleneck.register_forward_hook(lambda m, i, o: feats["g"].append(i[0].detach().clone()))
    h2 = model.att_bottleneck.register_forward_hook(lambda m, i, o: feats["a"].append(i[0].detach().clone()))
    with torch.no_grad():
        for loader in (HS.loader(q_pids, q_cams, HS.Q_SEED), HS.loader(g_pids, g_cams, HS.G_SEED)):
            for imgs, _, _, adj in loader:
                model(imgs, adj)
    h1.remove(), h2.remove()
    sd = calibrate_bnneck(sd, torch.cat(feats["g"]), torch.cat(feats["a"]))
    model.load_state_dict(sd)
    model.eval()
    out = {}
    for metric in ("cosine", "euclidean"):
        for sample in ("evenly", "dense"):
            drv.args.dist_metric = metric
            drv.args.test_sample = sample
            drv.args.re_rank = False
            if sample == "evenly":
                ql, gl = list(HS.loader(q_pids, q_cams, HS.Q_SEED)), list(HS.loader(g_pids, g_cams, HS.G_SEED))
            else:  # the dense samplers: batch = ONE tracklet of n clips (b = 1: the reference's view(n, 1, -1) pooling)
                ql, gl = list(HS.dense_loader(q_pids, q_cams, HS.Q_SEED)), list(HS.dense_loader(g_pids, g_cams, HS.G_SEED))
            distmat = drv.test(model, ql, gl, "avg", False, return_distmat=True)
            # what test() does with it (train_vidreid_xent_htri.py:531-542), through the reference's own evaluate_rank
            cmc, mAP = sys.modules["torchreid.metrics"].evaluate_rank(distmat, q_pids, g_pids, q_cams, g_cams, use_metric_mars=True)
            rank1 = cmc[0]
            if metric == "cosine" and sample == "evenly":   # ... and once literally: test()'s own return value
                r1, m2 = drv.test(model, ql, gl, "avg", False)
                assert r1 == rank1 and m2 == mAP
            tag = "%s_%s" % (metric, sample)
            out[tag + "_rank1"] = np.float64(rank1)
            out[tag + "_mAP"] = np.float64(mAP)
            out[tag + "_cmc"] = np.asarray(cmc, dtype=np.float64)
            out[tag + "_distmat"] = np.asarray(distmat, dtype=np.float32)
            print(tag, "Rank-1 %.4f mAP %.6f" % (rank1, mAP))
    out.update(q_pids=q_pids, q_cams=q_cams, g_pids=g_pids, g_cams=g_cams,
               g_mean=sd["global_bottleneck.running_mean"].numpy(), g_var=sd["global_bottleneck.running_var"].numpy(),
               a_mean=sd["att_bottleneck.running_mean"].numpy(), a_var=sd["att_bottleneck.running_var"].numpy())
    path = os.path.join(HERE, "test_harness.npz")
    np.savez_compressed(path, **out)
    print("wrote test_harness.npz %.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
