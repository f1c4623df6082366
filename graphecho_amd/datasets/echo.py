"""EchoNet-Dynamic -> raw (clip, LV mask) samples (reference datasets/echo.py:19-291).

Host side only: FileList.csv / VolumeTracings.csv parsing, the split filter, the ">= 2 traced frames" filter, Motion-JPEG
decode + grey conversion, clip selection at the traced frame and the tracing -> mask rasterisation
(skimage.draw.polygon restated in datasets/raster.py).  /255 and any resize / crop run on the GPU (graphecho_amd.data).
Decoder and rasteriser parity is unpinned (cv2 / scikit-image are not installed here; see formats.py, raster.py).
"""
import collections
import csv
import os

import numpy as np

from .formats import bgr_to_gray, read_avi_mjpeg
from .raster import polygon


def loadvideo(filename, grey=True):
    """-> uint8 (channels, frames, height, width), 1 channel when grey (echo.py:294-328)."""
    if not os.path.exists(filename):
        raise FileNotFoundError(filename)
    v = read_avi_mjpeg(filename)                                        # (F, H, W, 3) RGB
    v = bgr_to_gray(v)[..., None] if grey else v
    return np.ascontiguousarray(v.transpose(3, 0, 1, 2))


class EchoSet:
    """`root/FileList.csv`, `root/VolumeTracings.csv`, `root/Videos/*.avi`; split in {train, val, test, all}.

    Sample = (clip uint8 (1, H, W, T) -- channel first, time last, the layout the clip-fold kernel takes --,
    LV mask uint8 (H, W) of the traced frame (LargeTrace: the last traced frame, SmallTrace: the first), 0, index).
    Defaults are the reference's: length 8, period 1, max_length 8, grey, clip starting AT the traced frame.
    Deviation: a clip that would run past the end of the video is padded with black frames (the reference indexes past
    the array and raises, echo.py:268)."""

    def __init__(self, root, split="train", target_type="LargeTrace", length=8, period=1, max_length=8, grey=True):
        if target_type not in ("LargeTrace", "SmallTrace"):
            raise ValueError("target_type must be LargeTrace or SmallTrace (the two the trainers use)")
        self.root, self.split, self.target_type = root, split.upper(), target_type
        self.length, self.period, self.max_length, self.grey = length, period, max_length, grey
        self.class_values = (1,)
        with open(os.path.join(root, "FileList.csv"), newline="") as f:
            rows = list(csv.DictReader(f))
        if self.split != "ALL":
            rows = [r for r in rows if r["Split"] == self.split]        # echo.py:107-110 (exact match, as written)
        videos = set(os.listdir(os.path.join(root, "Videos")))
        names = [r["FileName"] for r in rows]
        names = [n if n in videos or os.path.splitext(n)[1] else n + ".avi" for n in names]   # the public CSV omits ".avi"
        missing = sorted(set(names) - videos)
        if missing:
            raise FileNotFoundError(os.path.join(root, "Videos", missing[0]))
        self.frames = collections.defaultdict(list)                    # video -> traced frame numbers, in file order
        self.trace = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(os.path.join(root, "VolumeTracings.csv"), newline="") as f:
            header = f.readline().strip().split(",")
            if header != ["FileName", "X1", "Y1", "X2", "Y2", "Frame"]:
                raise ValueError("VolumeTracings.csv: unexpected header " + ",".join(header))
            for line in f:
                if not line.strip():
                    continue
                fn, x1, y1, x2, y2, fr = line.strip().split(",")
                fn = fn if fn.endswith(".avi") else fn + ".avi"        # echo.py:136
                fr = int(fr)
                if fr not in self.trace[fn]:
                    self.frames[fn].append(fr)
                self.trace[fn][fr].append((float(x1), float(y1), float(x2), float(y2)))
        self.fnames = [n for n in names if len(self.frames[n]) >= 2]    # echo.py:149-152

    def __len__(self):
        return len(self.fnames)

    def mask_of(self, name, frame, shape):
        """Volume tracing -> LV mask (echo.py:237-246): the chords' left end points down, the right ones back up."""
        t = np.array(self.trace[name][frame])
        x1, y1, x2, y2 = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
        x = np.concatenate((x1[1:], np.flip(x2[1:])))
        y = np.concatenate((y1[1:], np.flip(y2[1:])))
        r, c = polygon(np.rint(y).astype(np.int64), np.rint(x).astype(np.int64), shape)
        mask = np.zeros(shape, np.uint8)
        mask[r, c] = 1
        return mask

    def __getitem__(self, index):
        name = self.fnames[index]
        video = loadvideo(os.path.join(self.root, "Videos", name), self.grey)          # (c, f, h, w) uint8
        c, f, h, w = video.shape
        length = f // self.period if self.length is None else self.length
        if self.max_length is not None:
            length = min(length, self.max_length)
        key = self.frames[name][-1] if self.target_type == "LargeTrace" else self.frames[name][0]
        idx = key + self.period * np.arange(length)
        clip = np.zeros((c, length, h, w), np.uint8)
        ok = idx < f
        clip[:, ok] = video[:, idx[ok]]
        return np.ascontiguousarray(clip.transpose(0, 2, 3, 1)), self.mask_of(name, key, (h, w)), 0, index


class EchoFrames:
    """Single-frame view of an EchoSet: (traced frame uint8 (1, H, W), LV mask uint8 (H, W), 0, index) -- the target-domain
    stream of the CAMUS -> EchoNet adaptation (train_camus_echo.py:216-218 feeds single frames to the FPN; the reference's
    own Echo class hands out 8-frame clips there)."""

    def __init__(self, echo_set):
        self.set = echo_set
        self.class_values = echo_set.class_values

    def __len__(self):
        return len(self.set)

    def __getitem__(self, index):
        clip, mask, a, b = self.set[index]
        return np.ascontiguousarray(clip[..., 0]), mask, a, b
