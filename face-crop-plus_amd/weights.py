"""State-dict specs, deterministic weight generator and ``.pth`` loading.

The three networks keep the reference's state_dict key names so that the real
checkpoints (``retinaface_detector.pth`` / ``bsrgan_x4_enhancer.pth`` /
``bise_parser.pth``, reference ``models/_layers.py:12-35``,
``retinaface.py:52``, ``rrdb.py:35``, ``bise.py:120``) load unchanged when a
user supplies them.  There is no network here, so tests / bench use
:func:`generate_state_dict`: seeded, Kaiming-scaled conv weights and
*non-trivial* BatchNorm statistics (default ``mean=0,var=1`` would hide
folding bugs).
"""
from __future__ import annotations

import os
import numpy as np

# kind tags
CONV_W, CONV_B, BN_W, BN_B, BN_RM, BN_RV, BN_NBT = range(7)

WEIGHTS_FILENAMES = {
    "retinaface": "retinaface_detector.pth",
    "rrdb": "bsrgan_x4_enhancer.pth",
    "bisenet": "bise_parser.pth",
}


def _conv(spec, name, cout, cin, k, bias=False):
    spec.append((name + ".weight", (cout, cin, k, k), CONV_W))
    if bias:
        spec.append((name + ".bias", (cout,), CONV_B))


def _bn(spec, name, c):
    spec.append((name + ".weight", (c,), BN_W))
    spec.append((name + ".bias", (c,), BN_B))
    spec.append((name + ".running_mean", (c,), BN_RM))
    spec.append((name + ".running_var", (c,), BN_RV))
    spec.append((name + ".num_batches_tracked", (), BN_NBT))


def retinaface_spec():
    """Keys/shapes of ``RetinaFace`` (reference retinaface.py:93-110; torchvision
    ResNet-50 v1.5 body through ``IntermediateLayerGetter``, so no avgpool/fc)."""
    s = []
    _conv(s, "body.conv1", 64, 3, 7)
    _bn(s, "body.bn1", 64)
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), 1):
        for b in range(blocks):
            p = f"body.layer{li}.{b}"
            _conv(s, p + ".conv1", planes, inplanes, 1)
            _bn(s, p + ".bn1", planes)
            _conv(s, p + ".conv2", planes, planes, 3)
            _bn(s, p + ".bn2", planes)
            _conv(s, p + ".conv3", planes * 4, planes, 1)
            _bn(s, p + ".bn3", planes * 4)
            if b == 0:
                _conv(s, p + ".downsample.0", planes * 4, inplanes, 1)
                _bn(s, p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    for i, cin in enumerate((512, 1024, 2048), 1):
        _conv(s, f"fpn.output{i}.0", 256, cin, 1)
        _bn(s, f"fpn.output{i}.1", 256)
    for i in (1, 2):
        _conv(s, f"fpn.merge{i}.0", 256, 256, 3)
        _bn(s, f"fpn.merge{i}.1", 256)
    for k in (1, 2, 3):
        for nm, co, ci in (("conv3X3", 128, 256), ("conv5X5_1", 64, 256),
                           ("conv5X5_2", 64, 64), ("conv7X7_2", 64, 64),
                           ("conv7x7_3", 64, 64)):
            _conv(s, f"ssh{k}.{nm}.0", co, ci, 3)
            _bn(s, f"ssh{k}.{nm}.1", co)
    for head, nout in (("ClassHead", 2), ("BboxHead", 4), ("LandmarkHead", 10)):
        for i in range(3):
            _conv(s, f"{head}.{i}.conv1x1", 2 * nout, 256, 1, bias=True)
    return s


def rrdb_spec():
    """Keys/shapes of ``RRDBNet`` (reference rrdb.py:53-62, _layers.py:168-200)."""
    s = []
    _conv(s, "conv_first", 64, 3, 3, bias=True)
    for t in range(23):
        for r in (1, 2, 3):
            p = f"RRDB_trunk.{t}.RDB{r}"
            for c in range(1, 5):
                _conv(s, f"{p}.conv{c}", 32, 64 + 32 * (c - 1), 3, bias=True)
            _conv(s, f"{p}.conv5", 64, 192, 3, bias=True)
    for nm in ("trunk_conv", "upconv1", "upconv2", "HRconv"):
        _conv(s, nm, 64, 64, 3, bias=True)
    _conv(s, "conv_last", 3, 64, 3, bias=True)
    return s


def bisenet_spec():
    """Keys/shapes of ``BiSeNet`` (reference bise.py:191-193, _layers.py:206-368)."""
    s = []
    _conv(s, "cp.resnet.conv1", 64, 3, 7)
    _bn(s, "cp.resnet.bn1", 64)
    cin = 64
    for li, cout in enumerate((64, 128, 256, 512), 1):
        for b in range(2):
            p = f"cp.resnet.layer{li}.{b}"
            _conv(s, p + ".conv1", cout, cin, 3)
            _bn(s, p + ".bn1", cout)
            _conv(s, p + ".conv2", cout, cout, 3)
            _bn(s, p + ".bn2", cout)
            if b == 0 and (cin != cout or li != 1):
                _conv(s, p + ".downsample.0", cout, cin, 1)
                _bn(s, p + ".downsample.1", cout)
            cin = cout
    for nm, ci in (("arm16", 256), ("arm32", 512)):
        _conv(s, f"cp.{nm}.conv.conv", 128, ci, 3)
        _bn(s, f"cp.{nm}.conv.bn", 128)
        _conv(s, f"cp.{nm}.conv_atten", 128, 128, 1)
        _bn(s, f"cp.{nm}.bn_atten", 128)
    for nm in ("conv_head32", "conv_head16"):
        _conv(s, f"cp.{nm}.conv", 128, 128, 3)
        _bn(s, f"cp.{nm}.bn", 128)
    _conv(s, "cp.conv_avg.conv", 128, 512, 1)
    _bn(s, "cp.conv_avg.bn", 128)
    _conv(s, "ffm.convblk.conv", 256, 256, 1)
    _bn(s, "ffm.convblk.bn", 256)
    _conv(s, "ffm.conv1", 64, 256, 1)
    _conv(s, "ffm.conv2", 256, 64, 1)
    _conv(s, "conv_out.conv.conv", 256, 256, 3)
    _bn(s, "conv_out.conv.bn", 256)
    _conv(s, "conv_out.conv_out", 19, 256, 1)
    return s


SPECS = {"retinaface": retinaface_spec, "rrdb": rrdb_spec, "bisenet": bisenet_spec}

# Offset added to the "face" logit bias of the generated RetinaFace ClassHead so
# that a realistic *minority* of the priors clears det_threshold=0.6 on i.i.d.
# noise images (calibrated with tools/calibrate_cls_bias.py).
GENERATED_CLS_BIAS_SHIFT = 0.1


def generate_state_dict(model: str, seed: int = 0, as_torch: bool = True):
    """Deterministic random-init state dict with reference key names.

    conv weights ~ N(0, gain/fan_in); BN gamma in U(0.6,1.2), beta N(0,0.1),
    running_mean N(0,0.1), running_var U(0.5,1.5).  Residual-branch tails are
    damped so activations stay O(1) through 50+ layers in fp32.
    """
    rng = np.random.default_rng(np.random.PCG64(0x5EED0000 + 7919 * seed
                                                + {"retinaface": 1, "rrdb": 2, "bisenet": 3}[model]))
    out = {}
    for name, shape, kind in SPECS[model]():
        if kind == CONV_W:
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 2.0
            if model == "rrdb":
                # dense blocks: LeakyReLU(0.2) and x5*0.2 residual scaling
                gain = 1.0
            if name.endswith("conv1x1.weight"):
                gain = 0.01   # head inputs have std ~10: keep regressions O(1)
            if "conv_out.conv_out" in name:
                gain = 1.0
            if name.startswith("conv_last"):
                gain = 0.0003  # keeps the x4 output inside (0,1) so the clamp/round tail is exercised
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(gain / fan_in))
        elif kind == CONV_B:
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.05)
            if name == "conv_last.bias":
                a = a + np.float32(0.5)
            if model == "retinaface" and name.startswith("ClassHead"):
                # channels are (anchor0: bg, face, anchor1: bg, face)
                a = a.copy()
                a[1::2] += np.float32(GENERATED_CLS_BIAS_SHIFT)
        elif kind == BN_W:
            lo, hi = (0.6, 1.2)
            if name.endswith("bn3.weight") or (model == "bisenet" and name.endswith("bn2.weight")):
                lo, hi = (0.25, 0.5)     # damp the residual branch
            a = rng.uniform(lo, hi, shape).astype(np.float32)
        elif kind == BN_B:
            a = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        elif kind == BN_RM:
            a = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        elif kind == BN_RV:
            a = rng.uniform(0.5, 1.5, shape).astype(np.float32)
            if model == "retinaface" and name == "body.bn1.running_var":
                # stem sees raw 0..255 pixels minus the channel means: conv1's
                # output variance is ~2*E[x^2] ~ 1.1e4 — normalise it like a
                # trained BN would so activations are O(1) from layer1 on
                a = a * np.float32(11200.0)
        else:
            a = np.array(1, dtype=np.int64)
        out[name] = a
    if as_torch:
        import torch
        out = {k: (torch.from_numpy(np.ascontiguousarray(v)) if v.ndim else torch.tensor(int(v))) for k, v in out.items()}
    return out


def trained_like_statistics(sd, seed: int = 0, bn_decades: float = 1.5, outlier_fraction: float = 1e-3, outlier_scale: float = 6.0):
    """Steps 1 and 2 of ``trained_like_retinaface`` on ANY of the three networks' state dicts (reference key names): every
    conv -> BatchNorm pair scaled per output channel by 10^U(-d, d) (weights, ``running_mean``, ``running_var`` x s^2: exactly
    function-preserving but for BatchNorm's eps) and a fraction of every conv filter's entries multiplied by ``outlier_scale``.
    BatchNorm partners by name: ``...convN`` -> ``...bnN`` (ResNet blocks, stems), ``....0`` -> ``....1`` (downsample, FPN, SSH),
    ``....conv`` -> ``....bn`` (BiSeNet's ConvBNReLU), ``...conv_atten`` -> ``...bn_atten`` (its attention modules).  RRDBNet has no
    BatchNorm: only the heavy tails apply.  Returns a new dict of torch tensors."""
    import torch
    rng = np.random.default_rng(0x7A1ED + seed)
    out = {k: torch.as_tensor(v).clone() for k, v in sd.items()}
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    convs = [k[:-len(".weight")] for k in out if k.endswith(".weight") and out[k].ndim == 4]
    bn_of = {}
    for conv in convs:
        head, _, tail = conv.rpartition(".")
        for cand in ((head + ".bn" + tail[4:]) if tail.startswith("conv") and tail[4:].isdigit() else None,
                     (head + ".1") if tail == "0" else None,
                     (head + ".bn") if tail == "conv" else None,
                     (head + ".bn_atten") if tail == "conv_atten" else None):
            if cand and (cand + ".running_var") in out and out[cand + ".running_var"].shape[0] == out[conv + ".weight"].shape[0]:
                bn_of[conv] = cand
                break
    for conv, bn in bn_of.items():
        w = out[conv + ".weight"]
        s = f32(10.0 ** rng.uniform(-bn_decades, bn_decades, w.shape[0]))
        out[conv + ".weight"] = w * s[:, None, None, None]
        out[bn + ".running_mean"] = out[bn + ".running_mean"] * s
        out[bn + ".running_var"] = out[bn + ".running_var"] * s * s
    for conv in convs:
        w = out[conv + ".weight"]
        hit = f32(rng.random(tuple(w.shape)) < outlier_fraction)
        out[conv + ".weight"] = w * (1 + (outlier_scale - 1) * hit)
    return out


def trained_like_retinaface(sd, seed: int = 0, stream_gain=(10.0, 30.0, 100.0, 10.0), uniform_gain: bool = False,
                           bn_decades: float = 1.5, outlier_fraction: float = 1e-3, outlier_scale: float = 6.0):
    """A RetinaFace state dict with the statistics of a TRAINED checkpoint, derived from ``sd`` (normally the generated one).

    The release checkpoint cannot be fetched here (no network), and what the generated weights lack is exactly what a
    trained one has: arbitrary scales.  Three reparametrisations, all expressed on the reference's own keys:

    1. every conv -> BatchNorm pair gets a per-output-channel scale s = 10^U(-d, d) (``bn_decades``): ``W[c] *= s``,
       ``running_mean *= s``, ``running_var *= s^2`` — ``running_var`` then spans ~1e-3 ... 1e3 and the un-normalised conv
       output the same decades, as in trained nets (exactly function-preserving but for BatchNorm's eps);
    2. a fraction of every conv filter's entries is multiplied by ``outlier_scale`` (heavy tails; changes the function a
       little, the oracle sees the same weights);
    3. the residual stream of ResNet layer L is scaled per channel by k_c in [1, stream_gain[L-1]] (all channels at the
       gain itself with ``uniform_gain``): its producers' (bn3, downsample.1) gamma and beta times k, every consumer's
       (the following conv1 / downsample.0 / fpn.output) input channel divided by k — exactly function-preserving for
       k > 0 since ReLU is positively homogeneous; the stream's activations reach gain x their former size.

    Steps 1 and 2 are ``trained_like_statistics`` (any of the three networks).  Returns a new dict of torch tensors (``sd`` is not modified)."""
    import torch
    out = trained_like_statistics(sd, seed, bn_decades, outlier_fraction, outlier_scale)
    rng = np.random.default_rng(0x57EA4 + seed)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    blocks = (3, 4, 6, 3)
    for li, gain in enumerate(stream_gain, 1):
        c = 256 * 2 ** (li - 1)
        k = f32(np.full(c, gain) if uniform_gain else gain ** rng.uniform(0.0, 1.0, c))
        if not uniform_gain:
            k[rng.integers(0, c, max(1, c // 20))] = float(gain)         # a few channels sit at the gain itself
        pre = f"body.layer{li}"
        for b in range(blocks[li - 1]):
            for nm in ("bn3",) + (("downsample.1",) if b == 0 else ()):
                out[f"{pre}.{b}.{nm}.weight"] = out[f"{pre}.{b}.{nm}.weight"] * k
                out[f"{pre}.{b}.{nm}.bias"] = out[f"{pre}.{b}.{nm}.bias"] * k
            if b > 0:
                out[f"{pre}.{b}.conv1.weight"] = out[f"{pre}.{b}.conv1.weight"] / k[None, :, None, None]
        consumers = [f"body.layer{li + 1}.0.conv1", f"body.layer{li + 1}.0.downsample.0"] if li < 4 else []
        if li >= 2:
            consumers.append(f"fpn.output{li - 1}.0")
        for nm in consumers:
            out[nm + ".weight"] = out[nm + ".weight"] / k[None, :, None, None]
    return out


URL_ROOT = "https://github.com/mantasu/face-crop-plus/releases/download/v1.0.0/"   # reference _layers.py:13


def checkpoint_dirs():
    """Where a real checkpoint is looked for, in order: $FCP_WEIGHTS_DIR, then torch.hub's checkpoint cache
    (``torch.hub.get_dir()`` honours $TORCH_HOME / $XDG_CACHE_HOME) — the directory the reference's
    ``torch.hub.load_state_dict_from_url`` (_layers.py:27-35) downloads into."""
    dirs = []
    if os.environ.get("FCP_WEIGHTS_DIR"):
        dirs.append(os.environ["FCP_WEIGHTS_DIR"])
    try:
        import torch
        dirs.append(os.path.join(torch.hub.get_dir(), "checkpoints"))
    except Exception:
        dirs.append(os.path.join(os.path.expanduser("~/.cache/torch/hub"), "checkpoints"))
    return dirs


def find_checkpoint(model: str):
    """Locate a real checkpoint on disk (no network access)."""
    fn = WEIGHTS_FILENAMES[model]
    for d in checkpoint_dirs():
        c = os.path.join(d, fn)
        if os.path.isfile(c):
            return c
    return None


def _fetch_checkpoint(model: str):
    """What the reference does when the file is not cached: download it from the release page
    (_layers.py:33-35).  Skipped with FCP_OFFLINE=1; any failure is reported by the caller."""
    import torch
    return torch.hub.load_state_dict_from_url(URL_ROOT + WEIGHTS_FILENAMES[model], map_location="cpu", progress=False)


def load_state_dict(model: str, source=None, seed: int = 0, device=None):
    """What the model objects call.  Single process (or ``source`` already a state dict in memory): ``load_local``.
    Inside an initialised ``torch.distributed`` group of more than one rank (north_star: "RCCL broadcast of weights +
    per-rank independent batches"): ONLY rank 0 resolves ``source`` — reads the checkpoint, or downloads it — and its
    tensors reach the other ranks as one flat broadcast (``dist.broadcast_state_dict``, RCCL when the group's backend is
    "nccl", on ``device``); ranks > 0 never touch the checkpoint directory or the network.  A failure on rank 0 is
    raised on every rank (nobody is left waiting in the collective).  ``FCP_BROADCAST_WEIGHTS=0`` restores per-rank
    loading."""
    from . import dist as D
    if isinstance(source, dict) or not D.is_dist() or os.environ.get("FCP_BROADCAST_WEIGHTS", "1") == "0":
        return load_local(model, source, seed)
    import torch
    import torch.distributed as dist
    rank, world = D.rank_world()
    if world == 1:
        return load_local(model, source, seed)
    sd, err = None, None
    if rank == 0:
        try:
            sd = load_local(model, source, seed)
        except Exception as e:                       # noqa: BLE001 - reported on every rank below
            err = f"{type(e).__name__}: {e}"
    if dist.get_backend() == "nccl":                 # RCCL moves device tensors only: this rank's GPU for objects and weights alike
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
    else:
        device = None                                # gloo (CPU tests; two ranks sharing one GPU): host tensors
    status = [err]
    dist.broadcast_object_list(status, src=0, device=device)
    if status[0] is not None:
        raise RuntimeError(f"{model}: rank 0 could not load the weights it was to broadcast: {status[0]}")
    names = [(n, tuple(shp)) for n, shp, _ in SPECS[model]()]
    if rank == 0:
        sd = {n: (sd[n] if torch.is_tensor(sd[n]) else torch.as_tensor(sd[n])) for n, _ in names}
    else:                                            # same keys and shapes everywhere: the spec is the skeleton
        sd = {n: (torch.zeros(shp, dtype=torch.float32) if not n.endswith("num_batches_tracked") else torch.tensor(0))
              for n, shp in names}
    return _validated(model, D.broadcast_state_dict(sd, device))


def load_local(model: str, source=None, seed: int = 0):
    """``source``: a state dict, a ``.pth`` path, the string ``"generated"`` (the seeded random-init generator:
    an explicit opt-in for tests / benchmarks — its detections are meaningless), or None.

    None means "the real weights", as in the reference: a checkpoint found in ``checkpoint_dirs()``, else the
    reference's download; when neither works this raises — it never silently falls back to random weights.
    ``FCP_WEIGHTS=generated`` in the environment turns None into "generated" (with a warning), for smoke runs of
    the CLI on boxes without the checkpoints."""
    import torch
    if isinstance(source, dict):
        sd = source
    elif source == "generated":
        sd = generate_state_dict(model, seed)
    elif isinstance(source, str):
        sd = torch.load(source, map_location="cpu")
    else:
        path = find_checkpoint(model)
        if path is not None:
            sd = torch.load(path, map_location="cpu")
        elif os.environ.get("FCP_WEIGHTS") == "generated":
            import warnings
            warnings.warn(f"{model}: FCP_WEIGHTS=generated — using seeded RANDOM weights; detections, crops and "
                          f"masks are meaningless (plumbing / benchmark use only)")
            sd = generate_state_dict(model, seed)
        else:
            err = "FCP_OFFLINE=1"
            if os.environ.get("FCP_OFFLINE") != "1":
                try:
                    return _validated(model, _fetch_checkpoint(model))
                except Exception as e:               # no network, HTTP error, disk full, ...
                    err = f"{type(e).__name__}: {e}"
            raise FileNotFoundError(
                f"{model}: checkpoint {WEIGHTS_FILENAMES[model]} not found in {checkpoint_dirs()} and the download "
                f"from {URL_ROOT} failed ({err}).  Put the file into $FCP_WEIGHTS_DIR or the torch hub cache, pass "
                f"weights={{'{model}': <path | state dict>}}, or opt in to random weights explicitly with "
                f"weights={{'{model}': 'generated'}} / FCP_WEIGHTS=generated.")
    return _validated(model, sd)


def _validated(model: str, sd):
    want = {n: tuple(s) for n, s, _ in SPECS[model]()}
    for k, shp in want.items():
        if k not in sd:
            raise KeyError(f"{model}: missing key {k}")
        if k.endswith("num_batches_tracked"):
            continue
        if tuple(sd[k].shape) != shp:
            raise ValueError(f"{model}: {k} has shape {tuple(sd[k].shape)}, expected {shp}")
    return sd
