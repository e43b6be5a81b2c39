"""Disk persistence of the tile tuner's picks (host logic; no GPU needed: the device section is stubbed)."""
import json

import pytest

PINNED_VOCAB_13 = "3db1988866f7e876"     # sha256(repr(vocabulary))[:16] at TABLE_VERSION 13


@pytest.fixture
def tuner(monkeypatch, tmp_path):
    from face_crop_plus_amd import engine as E
    monkeypatch.setenv("FCP_TUNE_CACHE", str(tmp_path / "tune.json"))
    monkeypatch.setattr(E.Autotune, "cache", {})
    monkeypatch.setattr(E.Autotune, "_disk_loaded", False)
    monkeypatch.setattr(E.Autotune, "_disk_section", "gfx950|256|abi0")
    return E.Autotune


def test_picks_round_trip_through_the_user_table(tuner, tmp_path, monkeypatch):
    k1 = (256, 256, 3, 3, 1, 102400, 0, False, False, 1, 1, 1, None, 128)
    k2 = (1024, 512, 1, 1, 1, 25600, 0, True, False, 1, 1, 1, (512, 2), 0)
    tuner.cache.update({k1: (256, 256, 2), k2: (128, 64)})
    assert tuner.save() == str(tmp_path / "tune.json")
    table = json.loads((tmp_path / "tune.json").read_text())
    assert list(table) == ["gfx950|256|abi0"] and len(table["gfx950|256|abi0"]) == 2
    tuner.cache.clear()
    monkeypatch.setattr(tuner, "_disk_loaded", False)
    tuner.ensure_loaded()
    assert tuner.cache[k1] == (256, 256, 2) and tuner.cache[k2] == (128, 64)
    # another device / ABI sees nothing of it, and keeps its own section when it saves
    tuner.cache.clear()
    monkeypatch.setattr(tuner, "_disk_loaded", False)
    monkeypatch.setattr(tuner, "_disk_section", "gfx950|256|abi1")
    tuner.ensure_loaded()
    assert k1 not in tuner.cache
    tuner.cache[k1] = (128, 128)
    tuner.save()
    table = json.loads((tmp_path / "tune.json").read_text())
    assert sorted(table) == ["gfx950|256|abi0", "gfx950|256|abi1"]


def test_damaged_or_disabled_table_is_harmless(tuner, tmp_path, monkeypatch):
    (tmp_path / "tune.json").write_text("{ not json")
    with pytest.warns(UserWarning, match="autotune table"):
        tuner.ensure_loaded()
    tuner.cache.clear()                          # whatever the shipped table held for this stub section: irrelevant here
    monkeypatch.setenv("FCP_TUNE_CACHE", "0")
    monkeypatch.setattr(tuner, "_disk_loaded", False)
    tuner.ensure_loaded()
    assert tuner.cache == {} and tuner.save() is None


def test_new_picks_are_saved_with_a_debounce_and_at_exit(tuner, tmp_path, monkeypatch):
    """A tuning pass meets dozens of new shapes in a row: the user's table is rewritten at most once every two seconds, and what
    is still unsaved goes out through the atexit hook (`_save_quietly`); a failing save never escapes."""
    import time
    path = tmp_path / "tune.json"
    monkeypatch.setattr(tuner, "_last_save", 0.0)
    tuner.cache[(1,)] = (128, 64)
    monkeypatch.setattr(tuner, "_disk_dirty", True)
    tuner._save_soon()                                     # first pick: saved at once
    assert json.loads(path.read_text())["gfx950|256|abi0"] == {"(1,)": [128, 64]}
    tuner.cache[(2,)] = (256, 256)
    monkeypatch.setattr(tuner, "_disk_dirty", True)
    tuner._save_soon()                                     # within two seconds: not yet
    assert "(2,)" not in json.loads(path.read_text())["gfx950|256|abi0"]
    tuner._save_quietly()                                  # what atexit runs
    assert "(2,)" in json.loads(path.read_text())["gfx950|256|abi0"] and tuner._disk_dirty is False
    monkeypatch.setattr(tuner, "_disk_dirty", True)
    monkeypatch.setattr(tuner, "save", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("dictionary changed size during iteration")))
    tuner._save_quietly()                                  # swallowed: tuning still works in-process


def test_autotune_off_still_reads_the_tables_and_tables_have_their_own_switch(tuner, tmp_path, monkeypatch):
    """FCP_AUTOTUNE=0 suppresses tuning launches only — a tuned shape keeps its tile (class docstring); ignoring the shipped /
    user tables is FCP_TUNE_TABLES=0 (``bench.py --no-autotune``)."""
    k = (64, 64, 3, 3, 1, 4096, 0, False, False, 1, 1, 1, None, 0)
    tuner.cache[k] = (128, 64)
    tuner.save()
    tuner.cache.clear()
    monkeypatch.setattr(tuner, "enabled", False)               # what FCP_AUTOTUNE=0 sets
    monkeypatch.setattr(tuner, "use_tables", True)
    monkeypatch.setattr(tuner, "_disk_loaded", False)
    tuner.ensure_loaded()
    assert tuner.cache.get(k) == (128, 64)
    tuner.cache.clear()
    monkeypatch.setattr(tuner, "use_tables", False)            # FCP_TUNE_TABLES=0
    monkeypatch.setattr(tuner, "_disk_loaded", False)
    tuner.ensure_loaded()
    assert k not in tuner.cache
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.path.insert(0, %r); from face_crop_plus_amd import engine as E; print(E.Autotune.enabled, E.Autotune.use_tables)" % root
    for env, want in (({"FCP_AUTOTUNE": "0"}, "False True"), ({"FCP_TUNE_TABLES": "0"}, "True False")):
        e = {k_: v for k_, v in os.environ.items() if k_ not in ("FCP_AUTOTUNE", "FCP_TUNE_TABLES")}
        out = subprocess.run([sys.executable, "-c", code], env={**e, **env}, capture_output=True, text=True, timeout=120)
        assert out.stdout.split("\n")[-2] == want, out.stdout + out.stderr


def test_tile_vocabulary_is_pinned_to_the_table_version():
    """The tuned tables store tile tuples; their meaning is fixed per ``Autotune.TABLE_VERSION``.  Whoever changes what
    ``tile_candidates`` offers (or the flag bit a third element carries) gets this failure and either bumps TABLE_VERSION
    and this pin (the shipped table must then be re-measured: tools/dump_autotune.py) or restores the vocabulary."""
    import hashlib
    from face_crop_plus_amd import engine as E
    vocab = []
    for cout in (8, 32, 64, 128, 160, 192, 256, 512, 2048):
        for halo in (False, True):
            for wide in (False, True):
                for big in (False, True):
                    vocab.append((cout, halo, wide, big, E.tile_candidates(cout, halo, wide, big, True)))
    digest = hashlib.sha256(repr(vocab).encode()).hexdigest()[:16]
    pinned = {13: PINNED_VOCAB_13}
    assert E.Autotune.TABLE_VERSION in pinned, "TABLE_VERSION changed: pin the new vocabulary digest here"
    assert digest == pinned[E.Autotune.TABLE_VERSION], \
        f"tile vocabulary changed (digest {digest}) without a TABLE_VERSION bump"
    # every pick the shipped table holds is a tuple this build still understands
    import ast, json, os
    shipped = os.path.join(os.path.dirname(os.path.abspath(E.__file__)), "tuned", "mi355x.json")
    known = {tuple(c) for v in vocab for c in v[4]}
    table = json.load(open(shipped))
    cur = [sec for sec in table if sec.endswith(f"abi{E.Autotune.TABLE_VERSION}")]
    assert cur, "the shipped table has no section for the current TABLE_VERSION"
    for sec in cur:
        for k, v in table[sec].items():
            key = ast.literal_eval(k)
            if len(key) == 14:                                    # conv picks (chain picks carry their own vocabulary)
                assert tuple(v) in known, (k, v)
