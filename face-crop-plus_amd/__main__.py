"""Command line of the MI355X path: the reference's flags (``__main__.py:117-226``)
mapped one-to-one onto ``Cropper``; ``-c/--config`` JSON supplies defaults; negative
thresholds mean "disabled" (None).  ``python -m face_crop_plus_amd -i DIR``.

Multi-GPU: launch with ``python -m torch.distributed.run --nproc-per-node N -m face_crop_plus_amd ...``;
rank 0 alone reads (or downloads) the checkpoints and broadcasts them over RCCL (weights.load_state_dict), then
each rank takes every N-th file batch (face_crop_plus_amd/dist.py) — no data-path collective.
"""
from __future__ import annotations

import argparse
import json
import os
import sys


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="face-crop-plus", description="Face crop / align / enhance / parse on MI355X")
    p.add_argument("-c", "--config", type=str, help="JSON file whose keys override the defaults below")
    p.add_argument("-i", "--input_dir", type=str, help="directory with input images")
    p.add_argument("-o", "--output-dir", type=str, default=None)
    p.add_argument("-cn", "--clean-names", action="store_true", help="copy the files to <input_dir>_temp under OS-safe names first (utils.clean_names)")
    p.add_argument("-ci", "--clean-names-inplace", action="store_true", help="rename the files to OS-safe names in place first")
    p.add_argument("-s", "--output-size", type=int, nargs="+", default=[256, 256])
    p.add_argument("-f", "--output-format", type=str, default=None)
    p.add_argument("-r", "--resize-size", type=int, nargs="+", default=[1024, 1024])
    p.add_argument("-ff", "--face-factor", type=float, default=0.65)
    p.add_argument("-st", "--strategy", type=str, default="largest")
    p.add_argument("-p", "--padding", type=str, default="constant")
    p.add_argument("-a", "--allow-skew", action="store_true")
    p.add_argument("-l", "--landmarks", type=str, default=None)
    p.add_argument("-ag", "--attr-groups", type=json.loads, default=None)
    p.add_argument("-mg", "--mask-groups", type=json.loads, default=None)
    p.add_argument("-dt", "--det-threshold", type=float, default=0.6)
    p.add_argument("-et", "--enh-threshold", type=float, default=-1)
    p.add_argument("-b", "--batch-size", type=int, default=8)
    p.add_argument("-n", "--num-processes", type=int, default=1)
    p.add_argument("-d", "--device", type=str, default="auto")
    return p


def parse_args(argv=None) -> dict:
    parser = build_parser()
    pre, _ = parser.parse_known_args(argv)
    if pre.config:
        with open(pre.config) as f:
            cfg = {k.replace("-", "_"): v for k, v in json.load(f).items()}
        parser.set_defaults(**cfg)
    args = vars(parser.parse_args(argv))
    args.pop("config")
    if args["input_dir"] is None:
        raise ValueError("Input directory must be specified.")        # __main__.py:231-232 of the reference
    for k in ("det_threshold", "enh_threshold"):
        if args[k] is not None and args[k] < 0:
            args[k] = None
    if args["device"] == "auto":
        args["device"] = f"cuda:{os.environ.get('LOCAL_RANK', '0')}"
    return args


def main(argv=None):
    kwargs = parse_args(argv)
    input_dir, output_dir = kwargs.pop("input_dir"), kwargs.pop("output_dir")
    needs_clean, is_inplace = kwargs.pop("clean_names"), kwargs.pop("clean_names_inplace")
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    temp_dir = None
    if needs_clean or is_inplace:                       # __main__.py:266-274 of the reference
        from .utils import clean_names
        if rank == 0:
            clean_names(input_dir=input_dir, output_dir=None if is_inplace else input_dir + "_temp")
        if needs_clean and not is_inplace:
            output_dir = input_dir + "_faces" if output_dir is None else output_dir
            input_dir = temp_dir = input_dir + "_temp"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        # RCCL ("nccl" IS RCCL on ROCm), one rank per GPU; FCP_DIST_BACKEND=gloo only for ranks that share a device (tests)
        dist.init_process_group(os.environ.get("FCP_DIST_BACKEND", "nccl"))
        dist.barrier()                                  # rank 0 has finished renaming / copying
    from .cropper import Cropper
    cropper = Cropper(**kwargs)
    cropper.process_dir(input_dir, output_dir)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if temp_dir is not None and rank == 0:
        import shutil
        shutil.rmtree(temp_dir)


if __name__ == "__main__":
    sys.exit(main())
