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
