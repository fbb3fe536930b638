"""The reference's own scripts driven against the drop-in (tests/golden/run_reference_scripts.py): build container only -
skipped wherever /root/reference is absent (the GPU box never sees the reference in any form)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/semantic-slam"), reason="the reference lives in the build container only")
def test_reference_visualize_and_train_scripts_run_on_the_drop_in(tmp_path):
    """visualize_matches_sequence.SequenceMatcher (ctor with the reference's YAML + a checkpoint dict, extract(path),
    match_with_quality) reproduces tests/golden/e2e.npz through the drop-in's classes, and train.SemanticSLAMTrainer runs its
    real constructor, one epoch (B = 4, backward through selector / grid_sample / refiner, optimizer step, validation) and
    save_checkpoint on them; the checkpoint it writes loads back into the visualize script; visualize_matches.MatchVisualizer and the
    four test/*.py evaluation classes (eval-mode backbone) run their own entry points on it, their matchers agreeing with the oracle's
    M2 / M4 / M5.  A subprocess: the reference's
    `models` / `data` / `losses` package names must not leak into this test session's sys.modules."""
    out = tmp_path / "record.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "run_reference_scripts.py"), "--json", str(out)],
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    rec = json.loads(out.read_text())
    v, t = rec["visualize_matches_sequence"], rec["train"]
    assert v["keypoints_equal_e2e_npz"] and v["intensities_equal_e2e_npz"] and v["match_pairs_equal_e2e_npz"] and v["matches"] > 0
    assert v["descriptors_max_abs_err"] < 1e-5
    assert t["parameter_tensors_moved"] == "24 / 24" and t["grad_abs_sum_selector"] > 0 and t["grad_abs_sum_refiner"] > 0
    assert t["then_visualize_matches_sequence_on_that_checkpoint"]["matches"] >= 0
    # the other callers of the path, each through its own constructor and entry point (the test/*.py classes in eval mode)
    o = t["other_callers"]
    assert o["visualize_matches.MatchVisualizer"]["equal_to_oracle_M2"] and o["test_tracking.TrackingTester"]["equal_to_oracle_M5"]
    assert o["test_descriptor_quality.DescriptorQualityTester"]["equal_to_oracle_M4"]
    assert 0.0 <= o["test_repeatability.RepeatabilityTester"]["mean_repeatability"] <= 1.0
    assert o["test_performance.PerformanceTester"]["total_ms_cpu_eager"] > 0
