"""CPU tests of the CLIP-score harness (sliders_amd/clip_score.py; the reference's acceptance tooling: eval-scripts/clip_score.py:24-72
over the folder layout of eval-scripts/generate_images_sd1.py:110-215).  No CLIP weights exist offline: the plumbing runs on a small
random-init CLIPModel; the direction arithmetic is checked on known tables."""
import os

import numpy as np
import pytest
import torch

from sliders_amd import clip_score
from sliders_amd.generate import build_parser, scale_folder


def _write_images(root, scales, cases, samples=2):
    from PIL import Image
    rng = np.random.RandomState(0)
    for s in scales:
        d = os.path.join(root, scale_folder(s))
        os.makedirs(d, exist_ok=True)
        for c in cases:
            for k in range(samples):
                img = (rng.rand(96, 80, 3) * 255).astype(np.uint8)
                img[..., 0] = np.clip(img[..., 0].astype(int) + int(40 * s), 0, 255)       # something that moves with the scale
                Image.fromarray(img).save(os.path.join(d, f"{c}_{k}.png"))
    os.makedirs(os.path.join(root, "all"), exist_ok=True)                                   # the reference's overview folder: skipped


def test_scale_folder_names_round_trip():
    for s in (-5, -2, -1, 0, 0.5, 1, 2, 5):
        assert clip_score.scale_of_folder(scale_folder(s)) == float(s)
    assert scale_folder(0.5) == "half" and clip_score.scale_of_folder("all") is None
    assert clip_score.sorted_nicely(["10_0.png", "9_0.png", "1_1.png"]) == ["1_1.png", "9_0.png", "10_0.png"]


def test_clip_score_table_over_the_reference_layout(tmp_path):
    import pandas as pd
    root = str(tmp_path / "age_slider")
    scales, cases = [-2, -1, 0, 0.5, 1, 2], [0, 3, 7]
    _write_images(root, scales, cases)
    csv = str(tmp_path / "prompts.csv")
    pd.DataFrame({"case_number": [0, 3, 7, 9], "prompt": ["a person"] * 4, "evaluation_seed": [1, 2, 3, 4]}).to_csv(csv, index=False)
    means, d = clip_score.main(["--im_path", root, "--prompt", " old person ", "--prompts_path", csv, "--synthetic_clip"])
    out = pd.read_csv(os.path.join(root, "clip_scores.csv"))
    cols = [c for c in out.columns if c.startswith("clip_")]
    assert sorted(cols) == sorted(f"clip_{str(s).replace('half', '0.5')}" for s in ("-2", "-1", "0", "0.5", "1", "2"))
    assert set(means) == {float(s) for s in scales} and all(np.isfinite(v) for v in means.values())
    assert out.loc[out.case_number == 9, cols].isna().all(axis=None)            # a case without images stays empty (clip_score.py:53)
    assert out.loc[out.case_number != 9, cols].notna().all(axis=None)
    # per-case value = mean over the case's samples of logits_per_image[0][0]
    sc = clip_score.ClipScorer(synthetic=True)
    from PIL import Image
    vals = [float(sc.score([Image.open(os.path.join(root, "1", f"3_{k}.png"))], "old person")[0]) for k in range(2)]
    assert abs(float(out.loc[out.case_number == 3, "clip_1"].iloc[0]) - np.mean(vals)) < 1e-5
    assert np.isfinite(d["slope"]) and d["n_scales"] == 6
    # from_case / till_case window
    _, means2 = clip_score.score_folders(root, "old person", csv, sc, from_case=3, till_case=3)
    assert set(means2) == set(means)


def test_preprocess_matches_clip_image_processor_defaults():
    from PIL import Image
    img = Image.fromarray((np.random.RandomState(1).rand(300, 200, 3) * 255).astype(np.uint8))
    x = clip_score.preprocess(img, 224)
    assert x.shape == (3, 224, 224) and torch.isfinite(x).all()
    try:
        from transformers import CLIPImageProcessor
        ref = CLIPImageProcessor()(images=img, return_tensors="pt")["pixel_values"][0]
    except Exception:
        pytest.skip("no CLIPImageProcessor backend in this environment")
    assert float((x - ref).abs().max()) < 0.05       # same pipeline (bicubic resize implementations differ by a rounding of uint8)


def test_direction_and_comparison():
    up = {-2.0: 20.1, -1.0: 21.0, 0.0: 22.2, 1.0: 23.9, 2.0: 25.0}
    d = clip_score.direction(up)
    assert d["sign"] == 1.0 and d["monotone_fraction"] == 1.0 and abs(d["slope"] - 1.27) < 0.02
    down = {k: -v for k, v in up.items()}
    assert clip_score.direction(down)["sign"] == -1.0 and clip_score.direction(down)["monotone_fraction"] == 0.0
    c = clip_score.compare_directions(up, {k: 2 * v + 1 for k, v in up.items()})
    assert c["same_direction"] == 1.0 and abs(c["slope_ratio"] - 0.5) < 1e-6 and c["per_scale_correlation"] > 0.999
    assert clip_score.compare_directions(up, down)["same_direction"] == 0.0
    assert np.isnan(clip_score.direction({0.0: 1.0})["slope"])


def test_generate_cli_takes_the_evaluation_csv():
    a = build_parser().parse_args(["--synthetic", "--prompts_path", "p.csv", "--num_samples", "2", "--from_case", "3", "--till_case", "9"])
    assert a.prompts_path == "p.csv" and a.num_samples == 2 and (a.from_case, a.till_case) == (3, 9)
    with pytest.raises(SystemExit):
        clip_score.ClipScorer()          # neither weights nor --synthetic_clip: refuses (there is no download here)
