"""checkpoint.resolve_checkpoint: how the hub ids of sample.py:26-30 turn into local diffusers-layout directories (CPU only)."""
import os

import pytest

from rich_text_to_image_amd.checkpoint import DEFAULT_REPO, resolve_checkpoint


def _mk(p):
    os.makedirs(os.path.join(p, "unet"))
    return str(p)


def test_directory_env_override_and_hub_cache(tmp_path, monkeypatch):
    for k in ("RTDIFF_SD_PATH", "RTDIFF_SDXL_PATH", "HUGGINGFACE_HUB_CACHE", "HF_HOME"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("HOME", str(tmp_path / "home"))
    d = _mk(tmp_path / "ckpt")
    assert resolve_checkpoint(d, "SD") == d                                   # a directory is taken as is
    with pytest.raises(FileNotFoundError) as e:
        resolve_checkpoint(None, "SD")
    assert DEFAULT_REPO["SD"] in str(e.value) and "RTDIFF_SD_PATH" in str(e.value)
    monkeypatch.setenv("RTDIFF_SD_PATH", d)
    assert resolve_checkpoint(None, "SD") == d                                # default id -> environment override
    assert resolve_checkpoint(DEFAULT_REPO["SD"], "SD") == d
    with pytest.raises(FileNotFoundError):
        resolve_checkpoint("someone/other-model", "SD")                       # the override is for the family default only
    # hub cache: newest snapshot that holds a unet/
    hub = tmp_path / "hf" / "hub"
    old = _mk(hub / "models--stabilityai--stable-diffusion-xl-base-1.0" / "snapshots" / "aaaa")
    new = _mk(hub / "models--stabilityai--stable-diffusion-xl-base-1.0" / "snapshots" / "bbbb")
    os.utime(old, (1, 1))
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf"))
    assert resolve_checkpoint("stabilityai/stable-diffusion-xl-base-1.0", "SDXL") == new
    monkeypatch.delenv("HF_HOME")
    monkeypatch.setenv("HUGGINGFACE_HUB_CACHE", str(hub))
    assert resolve_checkpoint(None, "SDXL") == new
