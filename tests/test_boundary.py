"""Drop-in boundary without a GPU: registry, factory kwargs, state-dict keys, return
conventions, error behaviour, import side effects, C-ABI exports."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from lp16 import LP16, LP_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make(**kw):
    from torchreid import models
    cfg = dict(num_classes=7, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1, pyramid_part=True,
               use_pose=True, learn_graph=True, consistent_loss=False, num_parts=3, bnneck=True)
    cfg.update(kw)
    return models.init_model("vmgn", **cfg)


def test_registry_and_unknown_model(tmp_path):
    from torchreid import models
    assert "vmgn" in models.get_names()
    with pytest.raises(KeyError):
        models.init_model("nope")
    make(save_dir=str(tmp_path))  # the factory's source file is copied next to the logs
    assert os.path.isfile(os.path.join(str(tmp_path), "vmgn.py"))


def test_state_dict_contract():
    m = make()
    sd = m.state_dict()
    assert len(sd) == 402
    assert sd["conv1.weight"].shape == (64, 3, 7, 7)
    assert sd["graph_layers.1.linear.weight"].shape == (2048, 2048)
    assert sd["global_classifier.weight"].shape == (7, 2048) and sd["att_classifier.weight"].shape == (7, 2048)
    counts = {p: sum(k.startswith(p + ".") for k in sd) for p in ("layer1", "layer2", "layer3", "layer4_1", "layer4_2")}
    assert counts == {"layer1": 60, "layer2": 78, "layer3": 114, "layer4_1": 60, "layer4_2": 60}
    assert not m.global_bottleneck.bias.requires_grad and not m.att_bottleneck.bias.requires_grad
    assert abs(sum(p.numel() for p in m.parameters()) - 46.88e6) < 0.03e6
    # layer4_2 starts as a copy of layer4_1
    assert torch.equal(sd["layer4_1.0.conv1.weight"], sd["layer4_2.0.conv1.weight"])
    assert m.layer4_1[0].conv2.stride == (1, 1)  # last stride hard-wired to 1


def test_factory_asserts():
    with pytest.raises(AssertionError):
        make(use_pose=False, learn_graph=False)
    with pytest.raises(AssertionError):
        make(num_split=3)


def test_eval_and_train_return_conventions():
    m = make(num_gb=1)
    x, adj = torch.randn(2, 5, 3, 64, 32), torch.ones(2, 35, 35)
    m.eval()
    with torch.no_grad():
        y = m(x, adj)
    assert y.shape == (2, 4096)
    m.train()
    outs, feats = m(x, adj)
    assert [tuple(o.shape) for o in outs] == [(2, 7)] * 2 and [tuple(f.shape) for f in feats] == [(2, 2048)] * 2
    m.loss = {"xent"}
    assert isinstance(m(x, adj), list)
    m.loss = {"htri"}
    with pytest.raises(KeyError):
        m(x, adj)
    mc = make(num_gb=1, consistent_loss=True)
    mc.train()
    outs, feats = mc(x, adj)
    assert len(outs) == 5 and len(feats) == 5
    sum(o.sum() for o in outs).backward()  # autograd reaches the backbone
    assert mc.conv1.weight.grad is not None


def test_leaf_module_hooks_fire_on_cpu_forward():
    m = make(num_gb=1).eval()
    seen = []
    hooks = [mod.register_forward_hook(lambda mod, i, o: seen.append(type(mod).__name__))
             for mod in m.modules() if len(list(mod.children())) == 0]
    with torch.no_grad():
        m(torch.randn(1, 4, 3, 64, 32), torch.ones(1, 28, 28))
    for h in hooks:
        h.remove()
    assert seen.count("Conv2d") == 63 and "Linear" in seen and "AdaptiveAvgPool3d" in seen


def test_metrics_contract_without_gpu():
    from torchreid import metrics
    q, g = torch.randn(4, 16), torch.randn(60, 16)
    with pytest.raises(ValueError):
        metrics.compute_distance_matrix(q, g, "l1")
    with pytest.raises(AssertionError):
        metrics.compute_distance_matrix(q.numpy(), g)
    with pytest.raises(AssertionError):
        metrics.compute_distance_matrix(q[0], g)
    with pytest.raises(AssertionError):   # every gallery sample shares identity AND camera with the query (rank.py:83)
        metrics.evaluate_rank(np.zeros((4, 60)), np.zeros(4), np.zeros(60), np.zeros(4), np.zeros(60), use_metric_cuhk03=True)
    acc = metrics.accuracy([torch.eye(4), torch.eye(4).flip(0)], torch.arange(4), topk=(1, 2))
    assert acc.shape == (2, 2) and acc[0, 0] == 1.0 and acc[1, 0] == 0.0


def test_samplers_star_import_binds_driver_names():
    ns = {}
    exec("from torchreid.samplers import *", ns)
    for name in ("np", "torch", "random", "copy", "RandomIdentitySampler", "RandomIdentitySamplerV1", "RandomSampler",
                 "SequentialSampler", "Sampler", "BatchSampler", "SubsetRandomSampler", "WeightedRandomSampler"):
        assert name in ns   # reference samplers.py:9 star-imports torch.utils.data.sampler; the driver evals names (:227)
    data = [(None, pid, 0) for pid in range(6) for _ in range(5)]
    s = ns["RandomIdentitySampler"](data, batch_size=8, num_instances=4)
    order = list(iter(s))
    assert len(order) % 8 == 0 and len(order) == len(s)
    for i in range(0, len(order), 4):
        assert len({data[j][1] for j in order[i:i + 4]}) == 1
    list(iter(s))
    assert len(s) == 24   # the constructor-time estimate: iterating does not change len(trainloader)
    # the driver's call shape for every train sampler: eval(name)(data, batch_size=..., num_instances=...)  (:227)
    for name in ("RandomSampler", "RandomIdentitySampler", "RandomIdentitySamplerV1"):
        smp = eval(name, ns)(data, batch_size=8, num_instances=4)
        assert {int(i) for i in iter(smp)} <= set(range(len(data))) and len(smp) > 0
    assert sorted(iter(ns["RandomSampler"](data, 8, 4))) == list(range(len(data)))
    v1 = ns["RandomIdentitySamplerV1"](data, 4)   # reference signature (data_source, num_instances=4, **kwargs)
    assert v1.num_instances == 4 and len(v1) == 24


def test_losses_contract():
    from torchreid import losses
    x = torch.randn(8, 32, requires_grad=True)
    y = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3])
    ce = losses.CrossEntropyLabelSmooth(4, use_gpu=False)
    logits = torch.randn(8, 4)
    smooth = 0.9 * torch.nn.functional.one_hot(y, 4).float() + 0.1 / 4
    assert torch.allclose(ce(logits, y), (-smooth * torch.log_softmax(logits, 1)).sum(1).mean(), atol=1e-6)
    val = losses.DeepSupervision(losses.TripletLoss(), [x, 2 * x], y)
    val.backward()
    assert x.grad is not None


def test_cabi_library_exports_every_declared_symbol():
    """The C-ABI library loads (no GPU needed) and exports exactly what include/agrl_hip.h declares."""
    from torchreid import _hip
    header = open(os.path.join(ROOT, "include", "agrl_hip.h")).read()
    declared = set(re.findall(r"\b(agrl_[a-z0-9_]+)\s*\(", header))
    assert declared >= {"agrl_conv2d_bn_act", "agrl_graph_propagate", "agrl_distmat", "agrl_rank_topk"}
    if not os.path.exists(_hip.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "agrl.pytorch_amd", "csrc"), "-j8"])
    lib = _hip.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert set(_hip.SIGNATURES) == declared - {"agrl_version", "agrl_last_error", "agrl_reload_options", "agrl_built_with_ablation", "agrl_lp16_is_f16"}
    assert lib.agrl_version() >= 100
    assert lib.agrl_built_with_ablation() == 0   # the shipped library has no switch that removes work from a kernel
    assert lib.agrl_reload_options() == 0
    # argument validation happens before any launch, so it is checkable without a GPU
    assert lib.agrl_distmat(None, None, None, None, None, 1, 1, 64, 1, 0, 0, None, 0, None) != 0
    assert b"null pointer" in lib.agrl_last_error()
    assert bool(lib.agrl_lp16_is_f16()) == (_hip.LP_NAME == "fp16")


def test_both_16_bit_builds_of_the_library_export_the_same_abi():
    """The same sources are built twice (csrc/Makefile): libagrl_hip.so stores 16-bit data as fp16, libagrl_hip_bf16.so as
    bfloat16. Both export every declared symbol and say which one they are; asking for one and loading the other is an error."""
    import ctypes
    from torchreid import _hip
    header = open(os.path.join(ROOT, "include", "agrl_hip.h")).read()
    declared = set(re.findall(r"\b(agrl_[a-z0-9_]+)\s*\(", header))
    libdir = os.path.join(ROOT, "agrl.pytorch_amd", "lib")
    for name, is_f16 in (("libagrl_hip.so", 1), ("libagrl_hip_bf16.so", 0)):
        path = os.path.join(libdir, name)
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "agrl.pytorch_amd", "csrc"), "-j8"])
        h = ctypes.CDLL(path)
        for sym in declared:
            assert hasattr(h, sym), (name, sym)
        assert h.agrl_lp16_is_f16() == is_f16, name
    other = "bf16" if _hip.LP_NAME == "fp16" else "fp16"
    code = ("import os, sys; sys.path[:0] = [%r, %r]; os.environ['AGRL_HIP_LP16'] = %r; os.environ['AGRL_HIP_LIB'] = %r\n"
            "from torchreid import _hip\n"
            "try:\n    _hip.lib(); print('loaded')\nexcept _hip.HipLibraryError as e:\n    print('refused:', e)\n"
            % (ROOT, os.path.join(ROOT, "agrl.pytorch_amd"), other, _hip.LIB_PATH))
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300).stdout.decode()
    assert "refused" in out and "stores 16-bit data as" in out, out
    from torchreid import hip_ops as ops
    with pytest.raises(ValueError, match="AGRL_HIP_LP16"):
        ops.check_precision(other)
    assert ops.check_precision(_hip.LP_NAME) == _hip.LP_NAME and not ops.is_lp16("bf16x3") and ops.is_lp16(_hip.LP_NAME)


def test_missing_library_fails_loudly(monkeypatch):
    from torchreid import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libagrl_hip.so")
    with pytest.raises(_hip.HipLibraryError):
        _hip.lib()
    assert not _hip.available()


@pytest.mark.skipif(not os.path.isdir("/root/reference/torchreid"), reason="reference tree only exists in the build container")
def test_reference_overlay_resolves_out_of_scope_modules():
    """With AGRL_REFERENCE_ROOT set, sub-modules this build does not provide fall through to the reference tree
    while models/metrics/losses stay ours (route A)."""
    code = (
        "import torchreid, torchreid.models, torchreid.metrics, torchreid.lr_scheduler, torchreid.optimizers;"
        "import torchreid.utils.avgmeter, torchreid.utils.reidtools;"
        "print(torchreid.models.__file__); print(torchreid.lr_scheduler.__file__);"
        "print(torchreid.utils.avgmeter.__file__); print(torchreid.utils.reidtools.__file__)"
    )
    env = dict(os.environ, AGRL_REFERENCE_ROOT="/root/reference", PYTHONPATH=os.path.join(ROOT, "agrl.pytorch_amd"))
    out = subprocess.check_output(["python", "-c", code], env=env, cwd="/tmp").decode().split()
    assert "agrl.pytorch_amd" in out[0] and out[1].startswith("/root/reference")
    assert out[2].startswith("/root/reference") and "agrl.pytorch_amd" in out[3]


@pytest.mark.skipif(not os.path.isfile("/root/reference/train_vidreid_xent_htri.py"), reason="reference tree only exists in the build container")
def test_driver_import_block_under_overlay():
    """INTEGRATION.md route A: the reference driver's own import block (train_vidreid_xent_htri.py:1-29, read from the
    reference at test time) executes with this package first on PYTHONPATH -- in particular the names it takes from the
    modules this build shadows (set_wd, cur_time, visualize_ranked_results, calc_splits, re_ranking, the samplers' star
    names). tensorboardX / torchvision / h5py are absent from this image and stubbed."""
    code = r'''
import sys, types
for name in ("tensorboardX", "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "h5py"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["tensorboardX"].SummaryWriter = object
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
def _stub_attr(n):
    if n.startswith("__"):
        raise AttributeError(n)
    return type(n, (), {})
sys.modules["torchvision.transforms"].__getattr__ = _stub_attr
for cls in ("ToPILImage", "Resize", "RandomHorizontalFlip", "ToTensor", "Normalize", "Compose"):
    setattr(sys.modules["torchvision.transforms"], cls, type(cls, (), {}))   # what the reference's transforms.py star-imports
import scipy.misc, sklearn.metrics._base      # APIs the reference's 2019 pins still had (scipy<1.2, scikit-learn<0.24)
scipy.misc.imsave = lambda *a, **k: None
sys.modules["sklearn.metrics.base"] = sklearn.metrics._base
src = open("/root/reference/train_vidreid_xent_htri.py").read().split("parser = argparse.ArgumentParser")[0]
ns = {}
exec(compile(src, "driver_imports", "exec"), ns)
import torchreid.utils.torchtools as tt, torchreid.utils.reidtools as rt, torchreid.samplers as sm
assert "agrl.pytorch_amd" in tt.__file__ and "agrl.pytorch_amd" in rt.__file__ and "agrl.pytorch_amd" in sm.__file__
for name in ("set_wd", "cur_time", "visualize_ranked_results", "calc_splits", "re_ranking", "compute_model_complexity",
             "save_checkpoint", "AverageMeter", "Logger", "init_optim", "np", "torch", "random", "RandomSampler",
             "RandomIdentitySampler", "TripletLoss", "CrossEntropyLabelSmooth", "DeepSupervision", "models", "metrics"):
    assert name in ns, name
print("ok")
'''
    env = dict(os.environ, AGRL_REFERENCE_ROOT="/root/reference", PYTHONPATH=os.path.join(ROOT, "agrl.pytorch_amd"))
    out = subprocess.check_output(["python", "-c", code], env=env, cwd="/tmp").decode()
    assert out.strip().endswith("ok")


def test_shadowed_util_modules_carry_the_reference_names(tmp_path):
    """torchtools / reidtools shadow the reference's modules of the same name: every public function those define
    (reference torchtools.py:10-141, reidtools.py:13-80) exists here and behaves."""
    from torchreid.utils import torchtools as tt, reidtools as rt
    for name in ("cur_time", "adjust_learning_rate", "set_bn_to_eval", "set_wd", "count_num_param", "flip_tensor",
                 "weights_init_kaiming", "weights_init_xavier", "weights_init_classifier", "mem_report"):
        assert callable(getattr(tt, name)), name
    lin = torch.nn.Linear(4, 3)
    opt = torch.optim.SGD(lin.parameters(), lr=0.1, weight_decay=5e-4)
    tt.set_wd(opt, 0)
    assert opt.param_groups[0]["weight_decay"] == 0
    tt.adjust_learning_rate(opt, 0.1, epoch=45, stepsize=20)
    assert abs(opt.param_groups[0]["lr"] - 0.001) < 1e-12
    bn = torch.nn.BatchNorm1d(3).train()
    bn.apply(tt.set_bn_to_eval)
    assert not bn.training
    assert abs(tt.count_num_param(lin) - 15e-6) < 1e-12
    assert torch.equal(tt.flip_tensor(torch.arange(6).view(2, 3), 1), torch.tensor([[2, 1, 0], [5, 4, 3]]))
    assert re.match(r"\d{4}-\d\d-\d\d \d\d:\d\d:\d\d$", tt.cur_time())
    # visualize_ranked_results: tracklets (tuples of frame paths) and single images; same id + same camera skipped
    def touch(*parts):
        path = tmp_path.joinpath(*parts)
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_bytes(b"x")
        return str(path)
    class DS:
        query = [((touch("q", "0007", "a.jpg"), touch("q", "0007", "b.jpg")), 7, 0)]
        gallery = [(touch("g", "x.jpg"), 7, 0), (touch("g", "y.jpg"), 7, 1), (touch("g", "z.jpg"), 3, 0)]
    out = tmp_path / "ranked"
    rt.visualize_ranked_results(np.array([[0.1, 0.3, 0.2]]), DS, save_dir=str(out), topk=2)
    qdir = out / "id0007_cam0"
    assert sorted(os.listdir(str(qdir))) == ["gallery_top001_name_z.jpg", "gallery_top002_name_y.jpg", "query_top000"]
    assert sorted(os.listdir(str(qdir / "query_top000"))) == ["a.jpg", "b.jpg"]


def test_host_side_helpers_without_gpu():
    """Host logic that needs no device: the per-thread split-precision switch and the clip pooling of the eval harness on
    CPU tensors (reference train_vidreid_xent_htri.py:471-476)."""
    from torchreid import hip_ops as ops, _hip
    from torchreid.evaluation import pool_clips
    assert ops._gemm_code(torch.float32) == _hip.F32
    with ops.f32_split():
        assert ops._gemm_code(torch.float32) == _hip.F32X3 and ops._gemm_code(LP_DTYPE) == _hip.LP16
        with ops.f32_split(False):
            assert ops._gemm_code(torch.float32) == _hip.F32
        assert ops._gemm_code(torch.float32) == _hip.F32X3
    assert ops._gemm_code(torch.float32) == _hip.F32
    f = torch.arange(24, dtype=torch.float32).view(6, 4)
    assert torch.equal(pool_clips(f, 1), f)
    assert torch.equal(pool_clips(f, 3, "avg"), torch.stack([f[0:3].mean(0), f[3:6].mean(0)]))
    assert torch.equal(pool_clips(f, 3, "max"), torch.stack([f[0:3].max(0)[0], f[3:6].max(0)[0]]))


def test_host_side_switches_are_cached_until_reload(monkeypatch):
    """Round-5 review (dispatch sprawl): the Python-side AGRL_HIP_* switches are read from the environment once, like the library's
    own, instead of once per Bottleneck of every forward; _hip.reload_options() drops the cache together with the library's."""
    from torchreid import _hip
    from torchreid import hip_ops as ops
    _hip.reload_options()
    monkeypatch.delenv("AGRL_HIP_FUSE_SEAM", raising=False)
    assert ops.switch_on("AGRL_HIP_FUSE_SEAM") and ops.seam_enabled()
    monkeypatch.setenv("AGRL_HIP_FUSE_SEAM", "0")
    assert ops.seam_enabled()                      # cached: the forward's dispatch does not touch os.environ
    _hip.reload_options()
    assert not ops.seam_enabled()
    monkeypatch.delenv("AGRL_HIP_FUSE_SEAM")
    _hip.reload_options()
    assert ops.seam_enabled()


def test_weight_packs_are_cached_per_precision(monkeypatch):
    """pack_weights packs once per (device, precision) and answers from the cache while no parameter / buffer changed -- for every
    precision mode (round 6: a loop variable in the fp16x3 pack once shadowed the cache key, and every forward re-packed: correct
    results at half the speed, visible only in the bench line). Runs without a GPU: the in-loop form of fp16x3 packs with torch ops only."""
    from unittest import mock
    import torch
    from recipe import recipe_state_dict
    from torchreid import hip_ops as ops
    from torchreid import models
    from torchreid.models import _vmgn_hip as V
    monkeypatch.setenv("AGRL_HIP_SPLIT16_PLANES", "0")   # (the plane packs go through the library's pack kernels: GPU only)
    ops._SWITCHES.clear()
    m = models.init_model("vmgn", num_classes=4, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                          pyramid_part=True, use_pose=True, learn_graph=True)
    m.load_state_dict(recipe_state_dict(m.state_dict(), seed=0))
    m.eval()
    dev = torch.device("cpu")
    try:
        with mock.patch("torch.cuda.current_device", return_value=0):
            for prec in ("fp32", "bf16x3", "fp16x3"):
                p1 = V.pack_weights(m, dev, prec)
                assert V.pack_weights(m, dev, prec) is p1, prec
                assert set(m._hip_packs) >= {(0, prec)} and all(isinstance(k, tuple) for k in m._hip_packs)
            with torch.no_grad():
                m.conv1.weight.mul_(1.0)                      # an in-place edit bumps the version: the pack is rebuilt
            assert V.pack_weights(m, dev, "fp16x3") is not p1
    finally:
        ops._SWITCHES.clear()


def test_split16_weight_packers_follow_their_layout_contract():
    """The host-side packers of the conforming mode against include/agrl_hip.h, element by element (no GPU: pack-time torch logic).
    split16_inloop_weights: every 32-value k-tile of a row becomes [hi(k 4c..4c+3, 16+4c..16+4c+3), c = 0..3 | lo in the same order];
    split16_plane_weights: [wh | wh 2^-11 | wl] per source, and with pair_first the first source per 128-channel slab [wh | wl | wh 2^-11];
    both on w 2^k with max |w| 2^k in [2^13, 2^14), hi + lo reproducing w 2^k to 2^-21 of the largest weight."""
    import numpy as np
    import torch
    from torchreid import hip_ops as ops
    g = torch.Generator().manual_seed(11)
    w = torch.randn((6, 64), generator=g) * 0.03
    w[0, :3] = torch.tensor([0.0, 1e-9, -2e-7])
    p = ops.split16_inloop_weights(w)
    k = round(np.log2(1.0 / p.agrl_unscale))
    ws = (w * 2.0 ** k).numpy()
    assert 2 ** 13 <= np.abs(ws).max() < 2 ** 14 and p.agrl_presplit and tuple(p.shape) == (6, 64) and p.dtype == torch.float32
    halves = p.numpy().view(np.float16).reshape(6, 2, 2, 4, 8)          # (row, tile, {hi, lo}, c, e)
    for row in range(6):
        for tile in range(2):
            for c in range(4):
                idx = [32 * tile + 4 * c + e for e in range(4)] + [32 * tile + 16 + 4 * c + e for e in range(4)]
                hi = ws[row, idx].astype(np.float16)
                lo = (ws[row, idx] - hi.astype(np.float32)).astype(np.float16)
                assert np.array_equal(halves[row, tile, 0, c], hi) and np.array_equal(halves[row, tile, 1, c], lo)
    assert torch.equal(ops.split16_true_weights(p), w)
    # plane weights: a two-source tensor [W1 (256) | W2 (128)], triples and the pair form of the first source
    w2 = torch.randn((4, 384), generator=g) * 0.05
    t, u = ops.split16_plane_weights(w2, segments=[256, 128])
    tp, up = ops.split16_plane_weights(w2, segments=[256, 128], pair_first=True)
    assert u == up and t.dtype == torch.float16 and tuple(t.shape) == tuple(tp.shape) == (4, 3 * 384)
    s2 = (w2 / u).numpy()
    wh = s2.astype(np.float16)
    wl = (s2 - wh.astype(np.float32)).astype(np.float16)
    whs = (wh.astype(np.float32) * 2.0 ** -11).astype(np.float16)
    tn, tpn = t.numpy(), tp.numpy()
    assert np.array_equal(tn[:, :256], wh[:, :256]) and np.array_equal(tn[:, 256:512], whs[:, :256]) and np.array_equal(tn[:, 512:768], wl[:, :256])
    assert np.array_equal(tn[:, 768:896], wh[:, 256:]) and np.array_equal(tn[:, 896:1024], whs[:, 256:]) and np.array_equal(tn[:, 1024:], wl[:, 256:])
    for c in range(2):      # first source, slab c: [wh_c | wl_c | wh_c 2^-11]
        base = 384 * c
        assert np.array_equal(tpn[:, base:base + 128], wh[:, 128 * c:128 * c + 128])
        assert np.array_equal(tpn[:, base + 128:base + 256], wl[:, 128 * c:128 * c + 128])
        assert np.array_equal(tpn[:, base + 256:base + 384], whs[:, 128 * c:128 * c + 128])
    assert np.array_equal(tpn[:, 768:], tn[:, 768:])       # the second source stays a triple
    assert np.abs(wh.astype(np.float64) + wl.astype(np.float64) - s2).max() <= 2.0 ** -21 * np.abs(s2).max()
