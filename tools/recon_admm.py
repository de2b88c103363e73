#!/usr/bin/env python3
"""
Caller-flow counterpart of the reference's ``scripts/recon/admm.py:22-154`` on the MI355X engine:

    load_data(psf, data, **preprocess)  ->  ADMM(psf, **config.admm)  ->  set_data  ->  apply(disp_iter, save, ...)
    ->  final_reconstruction.npy

The reference drives this through Hydra (``configs/recon/defaults.yaml``); Hydra is not part of the hot path, so the
same YAML layout and key names are read with PyYAML and overridden with ``key.sub=value`` arguments:

    python tools/recon_admm.py input.psf=psf.npy input.data=raw.npy admm.n_iter=100 preprocess.downsample=1
    python tools/recon_admm.py --config my.yaml save=out_dir

Inputs are ``.npy`` / ``.npz`` arrays (image decoding is in front of the accelerated path); everything from the raw
arrays onward -- background removal, normalisation, flips, gray conversion, the solver, the read-out -- runs on the
device.  Pinned by ``tests/golden/caller_flow.npz`` (the imported reference's ``load_data`` + ``ADMM(psf, **cfg)``).
"""
import argparse
import os
import sys
import time

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# key names and defaults of configs/recon/defaults.yaml:9-82 (the parts scripts/recon/admm.py reads)
DEFAULTS = yaml.safe_load("""
input:
  psf: null
  data: null
  dtype: float32
  original: null
  background: null
torch: True
torch_device: cuda
preprocess:
  normalize: True
  downsample: 4
  shape: null
  flip: False
  bayer: False
  blue_gain: null
  red_gain: null
  single_psf: False
  gray: False
  bg_pix: [5, 25]
display:
  disp: 50
  plot: False
  gamma: null
save: True
admm:
  n_iter: 5
  mu1: 1.0e-6
  mu2: 1.0e-5
  mu3: 4.0e-5
  tau: 0.0001
  denoiser: null
  unrolled: false
  checkpoint_fp: null
  pre_process_model:
    network: null
    depth: 2
  post_process_model:
    network: null
    depth: 2
""")


def merge(cfg, other):
    for k, v in other.items():
        if isinstance(v, dict) and isinstance(cfg.get(k), dict):
            merge(cfg[k], v)
        else:
            cfg[k] = v
    return cfg


def override(cfg, item):
    key, _, val = item.partition("=")
    node = cfg
    parts = key.split(".")
    for p in parts[:-1]:
        node = node.setdefault(p, {})
    node[parts[-1]] = yaml.safe_load(val)


def run(config, out_dir=None):
    """The body of scripts/recon/admm.py:22-130.  Returns (result, timings)."""
    import torch

    import lenslesspicam_amd as lpa
    from lenslesspicam_amd.prep import load_data

    pre = config["preprocess"]
    psf, data = load_data(
        psf_fp=config["input"]["psf"], data_fp=config["input"]["data"], background_fp=config["input"]["background"],
        dtype=config["input"]["dtype"], downsample=pre["downsample"], bayer=pre["bayer"], blue_gain=pre["blue_gain"],
        red_gain=pre["red_gain"], plot=False, flip=pre["flip"], gamma=config["display"]["gamma"], gray=pre["gray"],
        single_psf=pre["single_psf"], shape=pre["shape"], use_torch=config["torch"],
        torch_device=config["torch_device"], bg_pix=pre["bg_pix"], normalize=pre["normalize"], bgr_input=False)
    disp = config["display"]["disp"]
    if disp is not None and disp < 0:                      # scripts/recon/admm.py:53-55
        disp = None
    save = config["save"]
    if save is True:
        save = out_dir or os.getcwd()
    if config["admm"].get("unrolled"):
        raise NotImplementedError("unrolled checkpoints with pre/post-processor networks are outside the hot path "
                                  "(lenslesspicam_amd.UnrolledADMM runs the unrolled iterations themselves)")
    t0 = time.time()
    recon = lpa.ADMM(psf, **config["admm"])                # unknown keys are swallowed, like the reference
    recon.set_data(data)
    if config["torch"] and torch.cuda.is_available():
        torch.cuda.synchronize()
    setup_s = time.time() - t0
    t0 = time.time()
    res = recon.apply(disp_iter=disp, save=save, gamma=config["display"]["gamma"], plot=config["display"]["plot"])
    if config["torch"] and torch.cuda.is_available():
        torch.cuda.synchronize()
    proc_s = time.time() - t0
    # scripts/recon/admm.py:124-127: `res[0]` -- the (D,H,W,C) image when apply() returned (image, ax) (plot=True),
    # its first depth plane otherwise
    img = res[0]
    img = img.cpu().numpy() if isinstance(img, torch.Tensor) else img
    if save:
        os.makedirs(str(save), exist_ok=True)
        np.save(os.path.join(str(save), "final_reconstruction.npy"), img)       # scripts/recon/admm.py:148
    return img, {"setup_s": setup_s, "processing_s": proc_s}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", help="YAML file with the layout of configs/recon/defaults.yaml")
    ap.add_argument("overrides", nargs="*", help="key.sub=value")
    args = ap.parse_args()
    cfg = merge({}, DEFAULTS)
    if args.config:
        merge(cfg, yaml.safe_load(open(args.config)))
    for item in args.overrides:
        override(cfg, item)
    if not cfg["input"]["psf"] or not cfg["input"]["data"]:
        ap.error("input.psf=... and input.data=... (.npy / .npz) are required")
    img, tm = run(cfg)
    print(f"Setup time : {tm['setup_s']} s")
    print(f"Processing time : {tm['processing_s']} s")
    print(f"Reconstruction shape: {img.shape}")


if __name__ == "__main__":
    main()
