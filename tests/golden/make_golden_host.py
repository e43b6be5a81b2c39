"""Golden vectors for the host-side logic, generated from the REFERENCE's own code (build container only).

    python tests/golden/make_golden_host.py

``utils.py`` and ``cropper.py`` import cv2 at module level and cannot be imported here, but the functions pinned
below use numpy / json only.  This script parses the two files, compiles ONLY the named definitions out of their
syntax trees into a scratch namespace (numpy, json, os, re, collections, shutil provided; an identity stand-in
for ``unidecode.unidecode`` is NOT provided, so clean_names is exercised on ASCII names only) and records inputs +
outputs in ``host_logic.npz`` / ``host_clean_names.json``.  Nothing of the reference's text is written to the repo.
"""
from __future__ import annotations

import ast
import collections
import json
import os
import re
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/face_crop_plus"


def extract(path, names, ns):
    """Compile the module-level assignments / functions (or ``Class.method``) called ``names`` into ``ns``."""
    tree = ast.parse(open(path).read())
    picked = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            picked.append(node)
        elif isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in names for t in node.targets):
            picked.append(node)
        elif isinstance(node, ast.ClassDef):
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and f"{node.name}.{sub.name}" in names:
                    picked.append(sub)
    mod = ast.Module(body=picked, type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, "exec"), ns)
    return ns


class _Tqdm:                                     # clean_names wraps the listing in tqdm.tqdm when desc is given
    @staticmethod
    def tqdm(it, **kw):
        return it


def main():
    ns = {"np": np, "json": json, "os": os, "re": re, "collections": collections, "shutil": shutil, "tqdm": _Tqdm}
    extract(os.path.join(REF, "utils.py"),
            {"STANDARD_LANDMARKS_5", "parse_landmarks_file", "get_landmark_slices_5", "get_ldm_slices", "clean_names"}, ns)
    extract(os.path.join(REF, "cropper.py"), {"Cropper._init_landmarks_target"}, ns)
    out = {"standard_landmarks_5": ns["STANDARD_LANDMARKS_5"]}

    # _init_landmarks_target for several (output_size, face_factor)
    cfgs = [((256, 256), 0.65), ((112, 96), 0.65), ((512, 384), 0.5), ((48, 64), 0.8), ((1024, 1024), 1.0)]

    class Self:
        pass
    for i, (size, ff) in enumerate(cfgs):
        s = Self()
        s.output_size, s.face_factor, s.num_std_landmarks = size, ff, 5
        ns["_init_landmarks_target"](s)
        out[f"target_{i}_cfg"] = np.array([size[0], size[1], ff], np.float64)
        out[f"target_{i}"] = s.landmarks_target
    s = Self()
    s.output_size, s.face_factor, s.num_std_landmarks = (256, 256), 0.65, 7
    try:
        ns["_init_landmarks_target"](s)
        out["target_bad_raises"] = np.array("")
    except ValueError as e:
        out["target_bad_raises"] = np.array(str(e))

    # landmark slices
    for k in (5, 12, 17, 21, 29, 49, 68, 98, 106):
        sl = ns["get_ldm_slices"](5, k)
        out[f"slices_{k}"] = np.array([[x.start, x.stop] for x in sl])
    for bad in ((5, 7), (6, 68)):
        try:
            ns["get_ldm_slices"](*bad)
            out[f"slices_bad_{bad[0]}_{bad[1]}"] = np.array("")
        except ValueError as e:
            out[f"slices_bad_{bad[0]}_{bad[1]}"] = np.array(str(e))

    # landmark files: json, csv (header + comma), txt (space separated), single-row txt
    rng = np.random.default_rng(3)
    with tempfile.TemporaryDirectory() as d:
        lm = rng.uniform(0, 200, (3, 5, 2)).round(3)
        names = ["a.jpg", "b c.png", "d.jpeg"]
        files = {
            "lm.json": json.dumps({n: v.tolist() for n, v in zip(names, lm)}),
            "lm.csv": "name,x1,y1,x2,y2,x3,y3,x4,y4,x5,y5\n" + "\n".join(
                ",".join([n.replace(" ", "_")] + [str(x) for x in v.reshape(-1)]) for n, v in zip(names, lm)) + "\n",
            "lm.txt": "\n".join(" ".join([n.replace(" ", "_")] + [str(x) for x in v.reshape(-1)]) for n, v in zip(names, lm)) + "\n",
            "one.txt": "only.jpg " + " ".join(str(x) for x in lm[0].reshape(-1)) + "\n",
        }
        for fn, text in files.items():
            p = os.path.join(d, fn)
            open(p, "w").write(text)
            try:
                l, f = ns["parse_landmarks_file"](p)
                out[f"lmfile_{fn}_landmarks"] = np.asarray(l)
                out[f"lmfile_{fn}_names"] = np.asarray(f).astype(str)
            except Exception as e:                              # record how the reference itself behaves
                out[f"lmfile_{fn}_error"] = np.array(type(e).__name__)
            out[f"lmfile_{fn}_text"] = np.array(text)

        # clean_names on ASCII names (copy mode and in-place), default arguments
        src = os.path.join(d, "names")
        os.makedirs(src)
        raw = ["a?b#c.d.png", "dup.jpg", "DUP.jpg", "dup-1.jpg", "plain.png", "x" * 80 + ".jpg", "we!rd name;.jpeg", "Dup.JPG"]
        for i, n in enumerate(raw):
            open(os.path.join(src, n), "wb").write(bytes([i]))
        listing = os.listdir(src)                                # the reference iterates in os.listdir order
        dst = os.path.join(d, "clean")
        ns["clean_names"](src, dst, desc=None)
        copy_map = {open(os.path.join(dst, n), "rb").read()[0]: n for n in os.listdir(dst)}
        short = os.path.join(d, "short")
        ns["clean_names"](src, short, max_chars=len(src) + 20, desc=None)
        short_map = {open(os.path.join(short, n), "rb").read()[0]: n for n in os.listdir(short)}
        json.dump({"raw": raw, "listing_order": listing, "src_dir_len": len(src),
                   "copy": {str(k): v for k, v in copy_map.items()},
                   "short_max_chars_extra": 20, "short": {str(k): v for k, v in short_map.items()}},
                  open(os.path.join(HERE, "host_clean_names.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "host_logic.npz"), **out)
    print("wrote host_logic.npz", sorted(out)[:6], "...", len(out), "entries; host_clean_names.json")


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    main()
