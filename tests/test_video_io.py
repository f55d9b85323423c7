"""CPU: the video stage's on-disk contract (save_videos_grid / the metrics loader's split)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from neurons_amd.video_io import load_gif, make_grid, save_videos_grid, split_gt_pred, video_grid_frames  # noqa: E402


def test_make_grid_layout():
    x = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5) + 1
    g = make_grid(x, nrow=6)
    assert g.shape == (3, 4 + 2 * 2, 2 * (5 + 2) + 2)
    assert torch.equal(g[:, 2:6, 2:7], x[0]) and torch.equal(g[:, 2:6, 9:14], x[1])
    assert float(g[:, :2].abs().sum()) == 0 and float(g[:, :, 7:9].abs().sum()) == 0          # padding is pad_value 0
    g2 = make_grid(torch.cat([x, x, x]), nrow=4)                                             # 6 images, 4 per row -> 2 rows
    assert g2.shape == (3, 2 * (4 + 2) + 2, 4 * (5 + 2) + 2) and torch.equal(g2[:, 8:12, 2:7], x[0])
    assert torch.equal(make_grid(x[:1]), x[0])                                               # single image: returned as is
    assert make_grid(x[:, :1]).shape[0] == 3                                                 # grey -> 3 channels


def test_frames_truncate_like_the_reference():
    v = torch.full((2, 3, 1, 2, 2), 0.999)
    f = video_grid_frames(v)
    assert f[0].dtype == np.uint8 and f[0][2, 2, 0] == 254                                   # int(0.999 * 255) = 254, not 255
    f = video_grid_frames(v * 2 - 1, rescale=True)
    assert f[0][2, 2, 0] in (254, 253)


def test_gif_round_trip_and_split(tmp_path):
    t, h, w = 4, 16, 16
    gt = torch.zeros(1, 3, t, h, w)
    pred = torch.zeros(1, 3, t, h, w)
    for i in range(t):
        gt[0, 0, i] = (i + 1) / t            # red ramp over time on the left
        pred[0, 2, i] = 1.0 - i / t          # blue ramp on the right
    path = os.path.join(tmp_path, "out", "test1.gif")
    save_videos_grid(torch.cat([gt, pred]), path)
    gif = load_gif(path)
    assert gif.shape == (t, h + 4, 2 * (w + 2) + 2, 3)
    g, p = split_gt_pred(gif)
    assert g.shape == p.shape == (t, h + 4, w + 3, 3)
    for i in range(t):
        assert abs(int(g[i, 2 + h // 2, 2 + w // 2, 0]) - int((i + 1) / t * 255)) <= 8        # palette quantisation tolerance
        assert abs(int(p[i, 2 + h // 2, 1 + w // 2, 2]) - int((1.0 - i / t) * 255)) <= 8
        assert int(g[i, 2 + h // 2, 2 + w // 2, 2]) <= 8 and int(p[i, 2 + h // 2, 1 + w // 2, 0]) <= 8
