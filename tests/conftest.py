import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "f16x3_only: a feature of the fp16x3 path (split32 tensors, fused stem, halo / 256-row tiles): "
                                       "its exact-fp32 parametrisation does not exist and is deselected, not skipped")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# Collection order of the ``-m gpu`` run (the driver stops at the first failure, ``-x``): the parity evidence of the hot path
# (SURVEY.md 8a rows against the reference goldens / the oracle) first, then the full BASELINE configurations, the kernel
# sweeps, the boundary, the distributed legs — and everything that spawns threads, worker processes or subprocesses LAST,
# so that a host-I/O robustness test can never again hide a hot-path golden (round 5 lost 112 tests that way).
GPU_ORDER = [
    "test_retina_post_gpu",      # a4-a8: priors, decode, NMS indices bit-exact vs reference goldens (ties, IoU == 0.4)
    "test_retinaface_gpu",       # a2-a3: heads vs reference golden, HF ResNet pin
    "test_warp_gpu",             # a13-a14
    "test_parse_enhance_gpu",    # a9-a11, a15-a18: RRDB, BiSeNet labels bit-exact, groups
    "test_third_party_pins",     # cv2 / torchvision fixtures when present
    "test_selfcheck_gpu",        # fp16x3 guard
    "test_fullsize_gpu",         # configs[1], [2], [4] at BASELINE sizes
    "test_conv_gpu", "test_chain_gpu", "test_fuzz_gpu",
    "test_batch_gpu",            # f1
    "test_cabi_c_harness", "test_torch_ops",      # b
    "test_dist_gpu",             # e
    "test_flow_gpu", "test_cropper_gpu",          # f2-f4: host I/O, threads, worker processes
]
# inside a file: tests that kill / spawn processes or race threads go to the very end of the whole run
PROCESS_TESTS = ("test_process_dir_recovers_after_a_decode_worker_dies", "test_two_croppers_two_threads_one_device",
                 "test_process_dir_without_detector_copies_borrowed_images", "test_process_dir_pipeline_is_deterministic")


def gpu_order_key(nodeid: str):
    """(rank of the file, process-test flag): stable sort key; unknown files keep their place between parity and host I/O."""
    mod = os.path.splitext(os.path.basename(nodeid.split("::")[0]))[0]
    rank = GPU_ORDER.index(mod) if mod in GPU_ORDER else GPU_ORDER.index("test_batch_gpu")
    last = any(name in nodeid for name in PROCESS_TESTS)
    return (1 if last else 0, rank)


def pytest_collection_modifyitems(config, items):
    # `precision` is an autouse parametrised fixture of the conv tests: features that only exist on the fp16x3 path used to show up as
    # 49 skips per run; the combination is not a test, so it is deselected
    drop = [it for it in items if it.get_closest_marker("f16x3_only") is not None
            and getattr(getattr(it, "callspec", None), "params", {}).get("precision") == "f32"]
    if drop:
        gone = set(map(id, drop))
        items[:] = [it for it in items if id(it) not in gone]
        config.hook.pytest_deselected(items=drop)
    items.sort(key=lambda it: gpu_order_key(it.nodeid))          # stable: the order inside a file is kept
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def device():
    import torch
    return torch.device("cuda:0")
