"""CPU-side consistency of the full-schedule fixtures (no GPU): every committed oracle trajectory under tests/golden/fullschedule/ is a case of
oracle/make_fullsize_golden.py, records the iterations its case lists, carries the fingerprints of the seeded weights, and has a tolerance
entry in BOTH checkers that consume it (tests/test_fullschedule_gpu.py and bench.py's pixel_parity) - a trajectory without one would fail the
driver-run bench line with a KeyError instead of a number."""
import ast
import glob
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _literal(path, name):
    """The dict literal assigned to `name` at module level of a source file (without importing the file: bench.py / the GPU test need a GPU)."""
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == name for t in node.targets):
            return ast.literal_eval(node.value)
    raise AssertionError(f"{name} not found in {path}")


def test_every_committed_trajectory_has_a_case_and_tolerances_everywhere():
    from oracle import make_fullsize_golden as mg
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "fullschedule", "*.pt")))
    names = [os.path.splitext(os.path.basename(f))[0] for f in files]
    assert {"config1", "config2_50", "config3_50", "config3_unit", "config5_50"} <= set(names)
    tol = _literal(os.path.join(ROOT, "tests", "test_fullschedule_gpu.py"), "TOL")
    upd = _literal(os.path.join(ROOT, "tests", "test_fullschedule_gpu.py"), "UPDATE_TOL")
    pix = _literal(os.path.join(ROOT, "bench.py"), "PIXEL_TOL")
    for f, n in zip(files, names):
        assert n in mg.CASES, n
        assert n in tol and n in upd and n in pix, f"{n}: tolerance entry missing"
        assert len(tol[n]) == 4 and len(pix[n]) == 4
        # the two checkers state the same bounds: latents, PSNR, max |d|, update-relative
        assert (tol[n][0], tol[n][1], tol[n][3], upd[n]) == pix[n], (n, tol[n], upd[n], pix[n])
        g = torch.load(f, weights_only=False)
        c = mg.CASES[n]
        assert sorted(g["checkpoints"]) == sorted(c["checkpoints"]), n
        assert g["case"]["steps"] == c["steps"] and g["case"]["gs"] == c["gs"] and g["case"]["R"] == c["R"]
        assert len(g["unet_fingerprint"]) == 64 and len(g["vae_fingerprint"]) == 64
        hw = c["hw"]
        assert tuple(g["lat0"].shape) == (1, 4, hw, hw) and g["image_u8"].dtype == torch.uint8 and tuple(g["image_u8"].shape) == (8 * hw, 8 * hw, 3)
        assert all(torch.isfinite(v).all() for v in g["checkpoints"].values())


def test_unit_variance_case_starts_at_unit_variance_and_is_dominated_by_its_update():
    """config3_unit is the regime check of round 6: the same schedule as config3_50, latents handed over at 1 / init_noise_sigma."""
    a = torch.load(os.path.join(ROOT, "tests", "golden", "fullschedule", "config3_unit.pt"), weights_only=False)
    b = torch.load(os.path.join(ROOT, "tests", "golden", "fullschedule", "config3_50.pt"), weights_only=False)
    assert abs(a["lat0"].std().item() - 1.0) < 0.02 and b["lat0"].std().item() > 10.0
    assert torch.allclose(a["lat0"] * (b["lat0"].std() / a["lat0"].std()), b["lat0"], rtol=1e-4, atol=1e-4)          # the same noise, scaled
    ra = (a["checkpoints"][50] - a["lat0"]).norm() / a["checkpoints"][50].norm()
    rb = (b["checkpoints"][50] - b["lat0"]).norm() / b["checkpoints"][50].norm()
    assert ra > 0.9 and rb < 0.4, (ra, rb)
