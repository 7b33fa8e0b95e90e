"""CLIP-score harness over the images `sliders_amd.generate --prompts_path ...` writes (the acceptance the reference measures its sliders
by: eval-scripts/clip_score.py:24-72 over the folders of eval-scripts/generate_images_sd1.py:110-215 / generate_images_xl.py).

    python -m sliders_amd.clip_score --im_path images/age_slider --prompt "old person" --prompts_path prompts.csv \
        [--clip_path DIR_WITH_openai_clip-vit-base-patch32 | --synthetic_clip] [--from_case 0 --till_case 1000000]

Layout read (the reference's): `<im_path>/<scale>/<case_number>_<sample>.png`, one folder per slider scale (`half` = 0.5; folders whose
name contains `all` and csv files are skipped, clip_score.py:38-39).  For every scale folder the image-text logit of CLIP
(`logits_per_image[0][0]`, clip_score.py:62-63) of every image against `--prompt` is averaged per case into column `clip_<scale>` of a
copy of the prompts csv, written to `<im_path>/clip_scores.csv` (clip_score.py:66-72); the per-scale means are printed.

On top of the reference's table this module reports the DIRECTION the north-star acceptance speaks of ("trained slider weights reproduce
the reference's CLIP-score direction"): least-squares slope of the per-scale mean score against the scale, and whether the means are
monotone in the scale (`direction()`); `compare_directions()` puts two such tables (this engine's slider vs the reference's) side by side.

No CLIP weights exist offline: `--synthetic_clip` builds a small random-init `CLIPModel` (plumbing only - the scores mean nothing), and a
real run needs `--clip_path` pointing at a local copy of openai/clip-vit-base-patch32 (the reference downloads it, clip_score.py:24-25).
Host-side evaluation tooling: nothing here is on the training hot path, and it runs on CPU or GPU alike."""
from __future__ import annotations

import argparse
import os
import re
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def sorted_nicely(names: List[str]) -> List[str]:
    """natural sort of file names (clip_score.py:27-30)"""
    conv = lambda t: int(t) if t.isdigit() else t
    return sorted(names, key=lambda k: [conv(c) for c in re.split("([0-9]+)", k)])


def scale_of_folder(name: str) -> Optional[float]:
    """'-2' -> -2.0, 'half' -> 0.5 (generate_images_sd1.py:118-119 writes 0.5 as 'half'); None for folders that are not a scale"""
    n = name.replace("half", "0.5")
    try:
        return float(n)
    except ValueError:
        return None


def preprocess(img, size: int = 224) -> torch.Tensor:
    """CLIPImageProcessor's defaults restated (resize the short side to `size` bicubic, centre crop, scale to [0, 1], normalise) so that a
    synthetic run needs no processor files: PIL image -> [3][size][size] float32"""
    from PIL import Image
    img = img.convert("RGB")
    w, h = img.size
    s = size / min(w, h)
    img = img.resize((max(size, round(w * s)), max(size, round(h * s))), Image.BICUBIC)
    w, h = img.size
    l, t = (w - size) // 2, (h - size) // 2
    img = img.crop((l, t, l + size, t + size))
    x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
    return (x - torch.tensor(CLIP_MEAN).view(3, 1, 1)) / torch.tensor(CLIP_STD).view(3, 1, 1)


class ClipScorer:
    """image-text logit of a CLIPModel: `score(images, prompt)` = logits_per_image[:, 0] (clip_score.py:60-63)"""

    def __init__(self, clip_path: Optional[str] = None, synthetic: bool = False, device: str = "cpu", seed: int = 0):
        from transformers import CLIPConfig, CLIPModel
        self.device = torch.device(device)
        self.processor = None
        if synthetic:
            torch.manual_seed(seed)
            cfg = CLIPConfig(text_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                              vocab_size=1024, max_position_embeddings=77),
                             vision_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                                image_size=224, patch_size=32), projection_dim=32)
            self.model = CLIPModel(cfg)
            self.image_size = 224
        else:
            if not clip_path:
                raise SystemExit("clip_score: --clip_path (a local copy of openai/clip-vit-base-patch32; there is no network here) or "
                                 "--synthetic_clip (random-init CLIP: plumbing only) is required")
            from transformers import CLIPProcessor
            self.model = CLIPModel.from_pretrained(clip_path)
            self.processor = CLIPProcessor.from_pretrained(clip_path)
            self.image_size = self.model.config.vision_config.image_size
        self.model.eval().to(self.device)
        self.synthetic = synthetic

    def _tokens(self, prompt: str) -> Dict[str, torch.Tensor]:
        if self.processor is not None:
            t = self.processor(text=[prompt], return_tensors="pt", padding=True)
            return {"input_ids": t["input_ids"], "attention_mask": t["attention_mask"]}
        # synthetic model: a deterministic stand-in for the BPE vocabulary (byte hashes), <eos> = the largest id as in CLIP
        vocab = self.model.config.text_config.vocab_size
        ids = [vocab - 2] + [1 + (hash_byte % (vocab - 3)) for hash_byte in prompt.encode()][:75] + [vocab - 1]
        return {"input_ids": torch.tensor([ids]), "attention_mask": torch.ones(1, len(ids), dtype=torch.long)}

    @torch.no_grad()
    def score(self, images: list, prompt: str) -> torch.Tensor:
        if self.processor is not None:
            px = self.processor(images=images, return_tensors="pt")["pixel_values"]
        else:
            px = torch.stack([preprocess(im, self.image_size) for im in images])
        tok = {k: v.to(self.device) for k, v in self._tokens(prompt).items()}
        out = self.model(pixel_values=px.to(self.device), **tok)
        return out.logits_per_image[:, 0].float().cpu()


def score_folders(im_path: str, prompt: str, prompts_path: str, scorer: ClipScorer, from_case: int = 0, till_case: int = 1000000):
    """-> (DataFrame with one `clip_<scale>` column per scale folder, {scale: mean score}).  Same table as clip_score.py:41-72."""
    import pandas as pd
    from PIL import Image
    df = pd.read_csv(prompts_path)
    cases = set(int(c) for c in df["case_number"])
    folders = sorted(m for m in os.listdir(im_path) if "all" not in m and ".csv" not in m and os.path.isdir(os.path.join(im_path, m)))
    means: Dict[float, float] = {}
    for name in folders:
        scale = scale_of_folder(name)
        per_case: Dict[int, List[float]] = {}
        for image in sorted_nicely(os.listdir(os.path.join(im_path, name))):
            try:
                case = int(image.split("_")[0].replace(".png", ""))
            except ValueError:
                continue
            if case not in cases or not (from_case <= case <= till_case):
                continue
            with Image.open(os.path.join(im_path, name, image)) as im:
                per_case.setdefault(case, []).append(float(scorer.score([im], prompt)[0]))
        col = f"clip_{name.replace('half', '0.5')}"
        df[col] = np.nan
        for case, vals in per_case.items():
            df.loc[df["case_number"] == case, col] = float(np.mean(vals))
        m = float(df[col].mean()) if per_case else float("nan")
        print(f"{name}: mean CLIP score {m:.4f} over {sum(len(v) for v in per_case.values())} images")
        if scale is not None and per_case:
            means[scale] = m
    return df, means


def direction(means: Dict[float, float]) -> Dict[str, float]:
    """What the slider does to the attribute's CLIP score: slope of the per-scale mean against the scale (least squares), its sign, and
    the fraction of neighbouring scale pairs whose means are ordered like the scales (1.0 = monotone increasing, 0.0 = decreasing)."""
    xs = np.array(sorted(means), dtype=np.float64)
    ys = np.array([means[x] for x in xs], dtype=np.float64)
    if len(xs) < 2:
        return {"slope": float("nan"), "sign": 0.0, "monotone_fraction": float("nan"), "n_scales": float(len(xs))}
    slope = float(np.polyfit(xs, ys, 1)[0])
    up = float(np.mean(np.diff(ys) > 0))
    return {"slope": slope, "sign": float(np.sign(slope)), "monotone_fraction": up, "n_scales": float(len(xs))}


def compare_directions(ours: Dict[float, float], ref: Dict[float, float]) -> Dict[str, float]:
    """This engine's slider against the reference's on the same prompts / seeds / scales: same sign of the slope (the acceptance), the
    ratio of the slopes, and the correlation of the per-scale means."""
    a, b = direction(ours), direction(ref)
    common = sorted(set(ours) & set(ref))
    corr = float(np.corrcoef([ours[s] for s in common], [ref[s] for s in common])[0, 1]) if len(common) >= 3 else float("nan")
    return {"same_direction": float(a["sign"] == b["sign"] and a["sign"] != 0), "slope_ours": a["slope"], "slope_ref": b["slope"],
            "slope_ratio": a["slope"] / b["slope"] if b["slope"] else float("nan"), "per_scale_correlation": corr}


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="clipScore", description="CLIP score of slider images per scale folder, and the slider's direction")
    p.add_argument("--im_path", required=True, help="folder with one sub-folder of images per slider scale")
    p.add_argument("--prompt", required=True, help="prompt to score the images against (the attribute)")
    p.add_argument("--prompts_path", required=True, help="csv with a case_number column (the file the images were generated from)")
    p.add_argument("--device", default="cpu")
    p.add_argument("--from_case", type=int, default=0)
    p.add_argument("--till_case", type=int, default=1000000)
    p.add_argument("--clip_path", default=None, help="local directory of openai/clip-vit-base-patch32 (model + processor files)")
    p.add_argument("--synthetic_clip", action="store_true", help="random-init CLIP (plumbing test; the scores mean nothing)")
    p.add_argument("--ref_scores", default=None, help="clip_scores.csv of the reference's slider on the same csv: directions are compared")
    return p


def _means_of_csv(path: str) -> Dict[float, float]:
    import pandas as pd
    df = pd.read_csv(path)
    out = {}
    for c in df.columns:
        if c.startswith("clip_"):
            s = scale_of_folder(c[5:])
            if s is not None and df[c].notna().any():
                out[s] = float(df[c].mean())
    return out


def main(argv=None):
    a = build_parser().parse_args(argv)
    scorer = ClipScorer(a.clip_path, a.synthetic_clip, a.device)
    print(f"Eval against prompt: {a.prompt.strip()}" + ("  [SYNTHETIC CLIP: plumbing only]" if a.synthetic_clip else ""))
    df, means = score_folders(a.im_path, a.prompt.strip(), a.prompts_path, scorer, a.from_case, a.till_case)
    out = os.path.join(a.im_path, "clip_scores.csv")
    df.to_csv(out, index=False)
    d = direction(means)
    print(f"direction: slope {d['slope']:+.4f} CLIP logits per unit of slider scale over {int(d['n_scales'])} scales, "
          f"monotone fraction {d['monotone_fraction']:.2f}; table: {out}")
    if a.ref_scores:
        c = compare_directions(means, _means_of_csv(a.ref_scores))
        print("against the reference's slider: " + ", ".join(f"{k} {v:+.4f}" for k, v in c.items()))
    return means, d


if __name__ == "__main__":
    main()
