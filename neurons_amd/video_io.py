"""On-disk contract of the video stage — host mirror of ``save_videos_grid`` (animatediff/utils/util.py:61-74) and of the
loader the metrics stage uses (run_metrics.py:36-46: ``gt, pred = np.split(gif, 2, axis=2)``).

Pure host-side byte work, so it is plain Python/numpy.  ``torchvision.utils.make_grid`` and
``imageio`` are third-party and absent from /root/reference and from this image: ``make_grid`` is restated from its
documented algorithm (defaults padding=2, pad_value=0), frames are written with Pillow's GIF encoder (the one imageio's
GIF plugin drives).  Parity of the layout is tested; byte-level parity of the GIF palette is UNPINNED.
"""
import math
import os
from typing import List, Tuple

import numpy as np
import torch


def make_grid(tensor: torch.Tensor, nrow: int = 8, padding: int = 2, pad_value: float = 0.0) -> torch.Tensor:
    """torchvision.utils.make_grid for a [B][C][H][W] batch (normalize=False)."""
    if tensor.dim() != 4:
        raise ValueError("expected a [B][C][H][W] tensor")
    if tensor.size(1) == 1:
        tensor = torch.cat((tensor, tensor, tensor), 1)
    if tensor.size(0) == 1:
        return tensor.squeeze(0)
    nmaps = tensor.size(0)
    xmaps = min(nrow, nmaps)
    ymaps = int(math.ceil(float(nmaps) / xmaps))
    height, width = int(tensor.size(2) + padding), int(tensor.size(3) + padding)
    grid = tensor.new_full((tensor.size(1), height * ymaps + padding, width * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= nmaps:
                break
            grid[:, y * height + padding:y * height + padding + tensor.size(2),
                 x * width + padding:x * width + padding + tensor.size(3)] = tensor[k]
            k += 1
    return grid


def video_grid_frames(videos: torch.Tensor, rescale: bool = False, n_rows: int = 6) -> List[np.ndarray]:
    """util.py:62-71: (b, c, t, h, w) in [0, 1] -> t uint8 HWC frames; ``(x * 255).astype(uint8)`` truncates like the reference."""
    videos = videos.detach().cpu().float().permute(2, 0, 1, 3, 4)
    outputs = []
    for x in videos:
        x = make_grid(x, nrow=n_rows)
        x = x.permute(1, 2, 0)
        if rescale:
            x = (x + 1.0) / 2.0
        outputs.append((x * 255).numpy().astype(np.uint8))
    return outputs


def save_videos_grid(videos: torch.Tensor, path: str, rescale: bool = False, n_rows: int = 6, fps: int = 8):
    """Drop-in for ``animatediff.utils.util.save_videos_grid`` (called at scripts/neuroclips_video.py:316-318 with
    ``torch.cat([gt_video, sample])`` so every frame holds ground truth and reconstruction side by side)."""
    from PIL import Image
    frames = video_grid_frames(videos, rescale, n_rows)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    imgs = [Image.fromarray(f) for f in frames]
    imgs[0].save(path, save_all=True, append_images=imgs[1:], duration=int(round(1000.0 / fps)), loop=0)


def load_gif(path: str) -> np.ndarray:
    """``iio.imread(path, index=None)``: all frames as uint8 [t][h][w][3]."""
    from PIL import Image, ImageSequence
    with Image.open(path) as im:
        return np.stack([np.asarray(f.convert("RGB")) for f in ImageSequence.Iterator(im)])


def split_gt_pred(gif: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """run_metrics.py:41: ``gt, pred = np.split(gif, 2, axis=2)`` (left half = ground truth, right half = reconstruction)."""
    gt, pred = np.split(gif, 2, axis=2)
    return gt, pred
