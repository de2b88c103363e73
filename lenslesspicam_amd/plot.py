"""Minimal display helper for ``apply(plot=True/save=...)``; matplotlib is optional."""
import os

import numpy as np


def plot_image(img, ax=None, gamma=None, title=None, save=False, name=None, pause=None):
    import matplotlib

    if not os.environ.get("DISPLAY"):
        matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    img = np.asarray(img, dtype=np.float32)
    if img.ndim == 4:  # (D,H,W,C): show the first depth plane
        img = img[0]
    img = img / max(float(img.max()), 1e-12)
    if gamma and gamma > 1:
        img = img ** (1.0 / gamma)
    if ax is None:
        _, ax = plt.subplots()
    ax.imshow(np.clip(img.squeeze(), 0, 1), cmap="gray" if img.shape[-1] == 1 else None)
    if title:
        ax.set_title(title)
    if save and name:
        os.makedirs(str(save), exist_ok=True)
        ax.figure.savefig(os.path.join(str(save), name))
    if pause:
        plt.draw()
        plt.pause(pause)
    return ax
