"""Disk persistence of the tile tuner's picks (host logic; no GPU needed: the device section is stubbed)."""
import json

import pytest


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
