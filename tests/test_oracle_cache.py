"""The committed-oracle-output helper of the full-architecture GPU tests (tests/oracle_cache.py): a stored value is returned only when
the oracle sources, the weights and the inputs it was computed from are the ones at hand - otherwise the live oracle runs.  CPU."""
import os

import torch

import oracle_cache as oc


def test_cached_value_needs_matching_sources_weights_and_inputs(tmp_path, monkeypatch):
    sd = {"a.weight": torch.arange(6.).reshape(2, 3), "b.bias": torch.ones(4)}
    fp = oc.weights_fingerprint(sd)
    assert fp == oc.weights_fingerprint({k: v.clone() for k, v in sd.items()})
    assert fp != oc.weights_fingerprint({**sd, "b.bias": torch.ones(4) * 1.5})
    calls = []

    def compute():
        calls.append(1)
        return torch.ones(2), [torch.zeros(3), 4.0]
    inputs = [torch.ones(2), {"k": torch.zeros(1)}, 3, None]
    monkeypatch.setenv("ORACLE_CACHE_WRITE", str(tmp_path))
    v, hit = oc.cached("unit", fp, inputs, compute)
    assert not hit and len(calls) == 1 and os.path.exists(tmp_path / "unit.pt")
    monkeypatch.delenv("ORACLE_CACHE_WRITE")
    monkeypatch.setattr(oc, "DIR", str(tmp_path))
    v2, hit2 = oc.cached("unit", fp, inputs, compute)
    assert hit2 and len(calls) == 1 and torch.equal(v2[0], v[0]) and v2[1][1] == 4.0
    assert not oc.cached("unit", fp, [torch.ones(2) * 2] + inputs[1:], compute)[1]                   # other inputs
    assert not oc.cached("unit", oc.weights_fingerprint({**sd, "b.bias": torch.zeros(4)}), inputs, compute)[1]      # other weights
    monkeypatch.setattr(oc, "_src_hash", "an edited oracle")
    assert not oc.cached("unit", fp, inputs, compute)[1]                                            # other oracle sources
    assert len(calls) == 4


def test_nothing_is_committed_that_the_current_oracle_did_not_produce():
    """Every file under tests/golden/fullsize_oracle carries a tag; a file whose tag the current oracle sources cannot reproduce is only
    ever ignored at run time - this test keeps the directory from silently filling with such files."""
    if not os.path.isdir(oc.DIR):
        return
    for f in os.listdir(oc.DIR):
        d = torch.load(os.path.join(oc.DIR, f))
        assert set(d) == {"tag", "value"} and len(d["tag"]) == 64, f
