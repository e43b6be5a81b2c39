"""Third-party pin of the ResNet-50 BODY TOPOLOGY (reference retinaface.py:93-99 =
`torchvision.models.resnet50()` + `IntermediateLayerGetter(layer2, layer3, layer4)`).

torchvision is not in the build container, so the network fixtures of this repo run the reference's RetinaFace over a
ResNet-50 written in tests/golden/_ref_loader.py — and "a stride placed on conv1 instead of conv2 would pass every test"
(VERDICT r4).  What the image DOES carry is Hugging Face `transformers`, whose `ResNetModel` is an independently written
implementation of the same architecture (`microsoft/resnet-50` = ResNet-50 v1.5, weight-compatible with torchvision's by its
conversion script: `layer_type="bottleneck"`, `downsample_in_bottleneck=False` puts the stride on the 3x3 conv, shortcut =
1x1 / stride conv + BN, stem 7x7 / 2 + BN + ReLU + MaxPool(3, 2, 1)).  This script loads the build's generated `body.*`
weights into that model under the obvious key map, runs one seeded input through it and stores the three stage outputs the
reference's `IntermediateLayerGetter` returns (stages 2-4) in tests/golden/hf_resnet50.npz together with the library
version.  At generation time it also checks that the stand-in of _ref_loader.py and the CPU oracle reproduce those maps, so
every network fixture produced through the stand-in is covered by the pin.

    python tests/golden/make_golden_hf_resnet.py          # build container; writes the .npz (data only)

tests/test_oracle_retinaface.py holds the oracle to the file (CPU), tests/test_retinaface_gpu.py the HIP kernels.
It pins an independent implementation of the documented topology, not torchvision's own code: row a3's "body pinned
against the build's stand-in only" becomes "pinned against a third-party ResNet-50"; the torchvision door
(tools/make_cv2_fixture.py) stays open for the real thing.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def hf_key(k: str) -> str | None:
    """`body.*` key of the reference's state dict (torchvision naming) -> transformers.ResNetModel key."""
    p = k.split(".")
    assert p[0] == "body"
    tail = p[-1]
    if p[1] == "conv1":
        return "embedder.embedder.convolution.weight"
    if p[1] == "bn1":
        return f"embedder.embedder.normalization.{tail}"
    stage, blk = int(p[1][len("layer"):]) - 1, int(p[2])
    base = f"encoder.stages.{stage}.layers.{blk}"
    if p[3] == "downsample":
        return f"{base}.shortcut.{'convolution' if p[4] == '0' else 'normalization'}.{tail}"
    i = int(p[3][-1]) - 1                                  # conv1 / bn1 -> layer.0, conv2 / bn2 -> layer.1, conv3 / bn3 -> layer.2
    return f"{base}.layer.{i}.{'convolution' if p[3].startswith('conv') else 'normalization'}.{tail}"


def main():
    import transformers
    from transformers import ResNetConfig, ResNetModel
    from face_crop_plus_amd import weights
    from oracle import retinaface_ref as R
    sd = weights.generate_state_dict("retinaface")
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
                       layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)
    model = ResNetModel(cfg).eval()
    own = {hf_key(k): v for k, v in sd.items() if k.startswith("body.")}
    missing, unexpected = model.load_state_dict(own, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    x = torch.from_numpy(np.random.default_rng(500).normal(0, 50, (1, 3, 96, 136)).astype(np.float32))
    with torch.no_grad():
        hs = model(x, output_hidden_states=True).hidden_states          # (embedding, stage 1, 2, 3, 4)
        feats = [hs[2], hs[3], hs[4]]
        ora = R.body(x, sd)
        import _ref_loader as L                                        # the stand-in every reference-derived fixture runs on
        stand = L._ResNet50().eval()
        stand.load_state_dict({k[len("body."):]: v for k, v in sd.items() if k.startswith("body.")}, strict=False)
        t = stand.maxpool(stand.relu(stand.bn1(stand.conv1(x))))
        st = []
        for name in ("layer1", "layer2", "layer3", "layer4"):
            t = getattr(stand, name)(t)
            if name != "layer1":
                st.append(t)
    for k, (a, b, c) in enumerate(zip(feats, ora, st)):
        scale = float(a.abs().max())
        eo, es = float((a - b).abs().max()) / scale, float((a - c).abs().max()) / scale
        print(f"stage {k + 2}: shape {tuple(a.shape)}, max {scale:.3g}; oracle vs HF {eo:.2e}, _ref_loader stand-in vs HF {es:.2e}")
        assert eo < 1e-5 and es < 1e-5, "the build's ResNet-50 body and transformers' ResNetModel disagree"
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_resnet50.npz")
    np.savez_compressed(out, x=x.numpy(), transformers_version=np.array(transformers.__version__),
                        torch_version=np.array(torch.__version__),
                        config=np.array("ResNetConfig(layer_type='bottleneck', depths=[3,4,6,3], hidden_sizes=[256,512,1024,2048], "
                                        "downsample_in_first_stage=False, downsample_in_bottleneck=False)"),
                        **{f"feat{k + 1}": f.numpy() for k, f in enumerate(feats)})
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
