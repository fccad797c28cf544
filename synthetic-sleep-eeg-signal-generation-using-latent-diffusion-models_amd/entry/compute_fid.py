"""FID between real and generated 30-s windows on U-Sleep features -- counterpart of /root/reference/src/compute_fid.py:341-419.
`python -m eegldm.entry.compute_fid --params_path params.pt --path_test_ids ids/ids_shhs_test.csv --path_pre_processed ... --samples "dir/sample*.npy"`

Flow of the reference script: USleep(in_chans=2, sfreq=100, depth=12, n_classes=5, input_size_s=30) with trained weights
(`/project/params.pt`, compute_fid.py:367 -- not shipped with the reference; `--params_path` names them here), features of the test
split's windows (crop [36:-36], EEG channel duplicated, bottleneck activation), features of the generated `sample_{i}.npy` windows,
FIDMetric.  Differences, on purpose: the real windows' features are streamed batch by batch into fp64 device moments instead of being
concatenated on the host; ALL generated files are used (the script keeps the last 64 of its list, :405); the extractor runs in eval mode
unless `--batch_stats` asks for the script's literal behaviour (it never calls model.eval(), so BatchNorm normalises every batch by its
own statistics)."""
import argparse
import glob

import numpy as np
import torch

from ..metrics import FeatureMoments, USleep, fid_features, frechet_distance
from .common import WindowLoader


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--params_path", default=None, help="U-Sleep state_dict (.pt); omitted = random initialisation (smoke runs only)")
    p.add_argument("--path_test_ids", default=None); p.add_argument("--path_pre_processed", default=None)
    p.add_argument("--type_dataset", default="shhs"); p.add_argument("--synthetic_windows", type=int, default=0)
    p.add_argument("--samples", required=True, help="glob of generated windows, each (n, 1, 3000) float32 (sample_trials.py:166-170)")
    p.add_argument("--batch_size", type=int, default=256); p.add_argument("--seed", type=int, default=42)
    p.add_argument("--batch_stats", action="store_true", help="BatchNorm with batch statistics: the reference never calls model.eval() (compute_fid.py:357-372)")
    p.add_argument("--reference_literal", action="store_true",
                   help="the quantity the reference script prints: batch-statistics BatchNorm AND only the last 64 generated windows (its loop keeps the last batch, compute_fid.py:395-414)")
    return p.parse_args(argv)


def main(args):
    if getattr(args, "reference_literal", False):
        args.batch_stats = True
    torch.manual_seed(args.seed)
    model = USleep(in_chans=2, sfreq=100, depth=12, with_skip_connection=True, n_classes=5, input_size_s=30, apply_softmax=False)   # compute_fid.py:357-365
    if args.params_path:
        model.load_state_dict(torch.load(args.params_path, map_location="cpu"))
    model.eval()
    dim = model.channels[-1]
    real = FeatureMoments(dim, ctx=model.ctx)
    loader = WindowLoader(args.path_pre_processed, args.batch_size, args.synthetic_windows, seed=args.seed, shuffle=False,
                          path_ids=args.path_test_ids, dataset=args.type_dataset)
    for batch in loader:
        real.update(fid_features(model, batch["eeg"], batch_stats=args.batch_stats))
    files = sorted(glob.glob(args.samples))
    if not files:
        raise FileNotFoundError(f"no generated windows match {args.samples!r}")
    if getattr(args, "reference_literal", False):          # the last 64 generated windows only
        keep, n = [], 0
        for f in reversed(files):
            keep.append(f); n += int(np.load(f, mmap_mode="r").reshape(-1, 3000).shape[0])
            if n >= 64:
                break
        files = sorted(keep)
    fake = FeatureMoments(dim, ctx=model.ctx)
    pending = []
    for i, f in enumerate(files):
        pending.append(np.load(f).astype(np.float32).reshape(-1, 1, 3000))
        if sum(len(a) for a in pending) >= args.batch_size or i == len(files) - 1:
            fake.update(fid_features(model, torch.from_numpy(np.concatenate(pending, 0)), batch_stats=args.batch_stats)); pending = []
    fid = frechet_distance(*fake.finalize(), *real.finalize())
    mode = ("reference-literal: batch-statistics BatchNorm, last 64 generated windows" if getattr(args, "reference_literal", False)
            else ("batch-statistics BatchNorm" if args.batch_stats else "eval-mode BatchNorm (running statistics), all generated windows"))
    print(f"FID: {fid}  ({real.n} real windows, {fake.n} generated windows, {dim} features; mode: {mode})")
    return fid


if __name__ == "__main__":
    main(parse_args())
