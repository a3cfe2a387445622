"""The video-FILE branch of the front end (`fetch_video(path)` / `sample_video(path)`), executed.

The reference decodes a path with ``decord.VideoReader`` (qwen-vl-utils `vision_process.py:228-256`: ``len(vr)``,
``vr.get_avg_fps()``, ``vr.get_batch(idx).asnumpy()`` in THWC order, ``permute(0, 3, 1, 2)``).  ``decord`` (like av / cv2 /
torchvision) is absent from this image, so the branch was dead code through round 4.  Here a test-only stand-in module with exactly
that surface -- a reader over an in-memory uint8 array -- is injected through ``sys.modules`` and the path route must equal the
tensor route frame for frame, for several (total_frames, fps) cases that exercise ``smart_nframes`` / ``frame_indices``.
What this does NOT pin: the pixels a real ffmpeg decode produces (parity of the decode itself stays unpinned)."""
import sys
import types

import numpy as np
import pytest
import torch

from spacer_amd.qwen_vl_utils import vision_process as VP

STORE = {}          # path -> (uint8 array [T, H, W, C], fps)
CALLS = []


class _FakeBatch:
    def __init__(self, arr):
        self._arr = arr

    def asnumpy(self):
        return self._arr


class _FakeVideoReader:
    def __init__(self, path, *a, **k):
        self.frames, self.fps = STORE[path]
        CALLS.append(("open", path))

    def __len__(self):
        return self.frames.shape[0]

    def get_avg_fps(self):
        return self.fps

    def get_batch(self, idx):
        CALLS.append(("get_batch", list(idx)))
        return _FakeBatch(self.frames[np.asarray(idx, dtype=np.int64)])


@pytest.fixture()
def fake_decord(monkeypatch):
    mod = types.ModuleType("decord")
    mod.VideoReader = _FakeVideoReader
    monkeypatch.setitem(sys.modules, "decord", mod)
    STORE.clear()
    CALLS.clear()
    return mod


def _video(total, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (total, h, w, 3), generator=g, dtype=torch.uint8).numpy()        # THWC, as decord returns it


CASES = [   # (total_frames, source fps, H, W, extra keys of the content part)
    (120, 30.0, 96, 128, {}),                       # fps rule: 120 / 30 * 2 = 8 frames
    (37, 24.0, 70, 90, {"nframes": 6}),             # explicit frame count
    (900, 25.0, 64, 64, {"fps": 1.0, "max_frames": 16}),      # clipped by max_frames
    (5, 10.0, 56, 84, {}),                          # fewer frames than FPS_MIN_FRAMES: total rounded down to the frame factor
]


@pytest.mark.parametrize("total,fps,h,w,extra", CASES)
def test_path_route_equals_tensor_route(fake_decord, total, fps, h, w, extra):
    arr = _video(total, h, w, seed=total)
    STORE["/videos/a.mp4"] = (arr, fps)
    ele_path = {"type": "video", "video": "/videos/a.mp4", "max_pixels": 128 * 28 * 28, **extra}
    ele_tensor = {"type": "video", "video": torch.from_numpy(arr).permute(0, 3, 1, 2).contiguous(), "source_fps": fps,
                  "max_pixels": 128 * 28 * 28, **extra}
    n = VP.smart_nframes(ele_path, total, fps)
    want_idx = VP.frame_indices(total, n)
    # fetch_video: decoded + resized float frames and the sampled fps
    out_p, fps_p = VP.fetch_video(ele_path, return_video_sample_fps=True)
    out_t, fps_t = VP.fetch_video(ele_tensor, return_video_sample_fps=True)
    assert ("get_batch", want_idx) in CALLS, "the reader was not asked for linspace(0, total - 1, nframes).round()"
    assert out_p.shape[0] == n and out_p.shape == out_t.shape
    assert torch.equal(out_p, out_t)
    assert fps_p == pytest.approx(fps_t) and fps_p == pytest.approx(n / max(total, 1e-6) * fps)       # reference :252
    # sample_video: the uint8 frames the GPU front end takes, the target size and the fps
    sp = VP.sample_video(ele_path)
    st = VP.sample_video(ele_tensor)
    assert sp is not None
    assert torch.equal(sp[0], st[0]) and sp[1] == st[1] and sp[2] == pytest.approx(st[2])
    assert torch.equal(sp[0], torch.from_numpy(arr[np.asarray(want_idx)]).permute(0, 3, 1, 2))
    assert sp[0].is_contiguous() and sp[0].dtype == torch.uint8
    assert sp[1][0] % VP.IMAGE_FACTOR == 0 and sp[1][1] % VP.IMAGE_FACTOR == 0


def test_process_vision_info_reads_a_path(fake_decord):
    arr = _video(48, 84, 112, seed=7)
    STORE["clip.mp4"] = (arr, 12.0)
    conv = [{"role": "user", "content": [{"type": "video", "video": "clip.mp4", "max_pixels": 64 * 28 * 28}, {"type": "text", "text": "?"}]}]
    images, videos, kw = VP.process_vision_info(conv, return_video_kwargs=True)
    assert images is None and len(videos) == 1 and videos[0].dim() == 4 and videos[0].shape[1] == 3
    assert kw["fps"][0] == pytest.approx(videos[0].shape[0] / 48 * 12.0)
    assert CALLS[0] == ("open", "clip.mp4")


def test_start_end_keys_raise_like_the_reference(fake_decord):
    """Reference :246-247: the decord reader refuses video_start / video_end."""
    STORE["a.mp4"] = (_video(16, 56, 56, 1), 8.0)
    for key in ("video_start", "video_end"):
        with pytest.raises(NotImplementedError):
            VP.fetch_video({"video": "a.mp4", key: 1.0})
        with pytest.raises(NotImplementedError):
            VP.sample_video({"video": "a.mp4", key: 1.0})


def test_missing_decord_is_loud(monkeypatch):
    monkeypatch.setitem(sys.modules, "decord", None)            # import decord -> ImportError
    with pytest.raises(ImportError, match="decord"):
        VP.fetch_video({"video": "whatever.mp4"})
    assert VP.sample_video({"video": "whatever.mp4"}) is None   # the GPU front end falls back to fetch_video, which raises
