"""weights.trained_like_retinaface (the real-checkpoint rehearsal's weight source) on the CPU oracle."""
import torch

from oracle import retinaface_ref as R


def test_stream_rescaling_is_function_preserving_and_statistics_are_trained_like():
    from face_crop_plus_amd import weights
    sd = weights.generate_state_dict("retinaface")
    x = R.preprocess(torch.randint(0, 256, (1, 3, 96, 128), generator=torch.Generator().manual_seed(1)).float())
    with torch.no_grad():
        base = R.forward_raw(x, sd)
        same = R.forward_raw(x, weights.trained_like_retinaface(sd, 0, bn_decades=0.0, outlier_fraction=0.0))
        for a, b in zip(base, same):                      # residual-stream scaling alone: the same function
            assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(a.abs().max()))
        t = weights.trained_like_retinaface(sd, 0, stream_gain=(10.0, 30.0, 500.0, 10.0))
        streams = R.body(x, t)
        streams = list(streams.values()) if isinstance(streams, dict) else list(streams)
        base_streams = R.body(x, sd)
        base_streams = list(base_streams.values()) if isinstance(base_streams, dict) else list(base_streams)
    assert set(t) == set(sd) and all(t[k].shape == sd[k].shape and t[k].dtype == sd[k].dtype for k in sd)
    rv = torch.cat([v.flatten() for k, v in t.items() if k.endswith("running_var") and k != "body.bn1.running_var"])
    assert float(rv.min()) < 2e-3 and float(rv.max()) > 5e2            # six decades of BatchNorm variance
    gains = [float(s.abs().max()) / float(b.abs().max()) for s, b in zip(streams, base_streams)]
    assert gains[1] > 50 and 3e3 < float(streams[1].abs().max()) < 32768          # layer3's stream: ~1e4, inside binary16's range
    assert all(sd[k].equal(weights.generate_state_dict("retinaface")[k]) for k in ("body.conv1.weight", "body.layer3.2.bn3.weight"))
