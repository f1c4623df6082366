"""CardiacUDA index + NIfTI volumes -> raw frames / clips and label maps (reference datasets/cardiac_uda.py:40-246).

Host side only: site filter over the `infos` index, the 90/10 id split, annotated-frame selection, clip sampling and
the NIfTI decode.  Resize / crop / /255 / one-hot / clip fold run on the GPU (graphecho_amd.data).  The reference seeds
the process-global `random` at import (cardiac_uda.py:35); this class owns a `random.Random(7777)` instead so that two
datasets do not perturb each other's draws -- the sequence of decisions per dataset is otherwise the reference's.
"""
import random

import numpy as np

from .formats import read_nifti
from .raster import fill_poly

ORGAN_NUM = {"1": 2, "2": 1, "3": 2, "4": 4}       # cardiac_uda.py:224

# one-hot planes per echo view (cardiac_uda.py:128-148): background first, then the chambers annotated in that view
VIEW_CLASS_VALUES = {"1": (0, 1, 2), "2": (0, 1), "3": (0, 1, 2), "4": (0, 1, 2, 3, 4)}


class CardiacUDASet:
    """infos: {id: {'dataset_name', 'views_images': {view: path|None}, 'views_labels': {view: path|None}}}
    (the dict stored in infos.npy).  Sample = (frames, label map, mask_index, index):
      single_frame: frames uint8 (1, H, W), labels uint8 (H, W);
      clips:        frames uint8 (1, H, W, T), labels uint8 (H, W, T), T = clip_length.
    """

    def __init__(self, infos, root, is_train, repeat=1, data_list=None, set_select=("Site_G",), view_num=("2",),
                 single_frame=True, total_length=40, clip_length=8, seg_parts=True, fill_mask=False, rng=None):
        self.fill_mask = fill_mask         # clips only: contour label maps -> filled masks (cardiac_uda.py:111-112)
        self.root, self.is_train, self.repeat = root, is_train, repeat
        self.set_select, self.view_num = tuple(set_select), tuple(view_num)
        self.single_frame, self.total_length, self.clip_length = single_frame, total_length, clip_length
        self.seg_parts = seg_parts
        self.class_values = VIEW_CLASS_VALUES[self.view_num[0]] if seg_parts else None
        self.rng = rng or random.Random(7777)
        self.data_dict = {k: {"images": v["views_images"], "masks": v["views_labels"]}
                          for k, v in infos.items() if v["dataset_name"] in self.set_select}   # get_dict :179-189
        self.id_list = list(self.data_dict)
        self.test_list = self.valid_list = None
        if is_train:                                                      # cardiac_uda.py:58-63
            self.train_list = self.rng.sample(self.id_list, int(len(self.id_list) * 0.9))
            self.valid_list = self.rng.sample(self.train_list, int(len(self.train_list) * 0.1))
            self.test_list = list(set(self.id_list).difference(self.train_list))
            self.id_list = self.train_list
        elif data_list is not None:
            self.id_list = list(data_list)
        self.num_data = len(self.id_list)

    def __len__(self):
        return self.num_data * self.repeat if self.is_train else self.num_data

    def input_select(self, images, masks):
        """Pick an annotated frame (> 100 labelled pixels) or a clip around one (cardiac_uda.py:191-216)."""
        if masks.ndim != 3:
            if self.single_frame:
                return images, masks, 0
            tile = lambda a: np.tile(a, (self.clip_length, 1, 1)).transpose(1, 2, 0)
            return tile(images), tile(masks), 0
        annotated = np.argwhere(np.sum(masks, axis=(0, 1)) > 100)
        if annotated.size == 0:
            return None, None, None
        select = int(self.rng.choice(list(annotated))[0])
        if self.single_frame:
            return images[:, :, select], masks[:, :, select], select
        if masks.shape[-1] == 3:
            return (np.tile(images[:, :, 1:2], (1, 1, self.clip_length)),
                    np.tile(masks[:, :, 1:2], (1, 1, self.clip_length)), np.array([select]))
        r = self.rng.randint(0, select if select < self.clip_length - 1 else self.clip_length - 1)
        start = select - r
        end = start + self.clip_length - 1
        return images[:, :, start:end], masks[:, :, start:end], np.array([r])

    def contour_to_mask(self, contours):
        """Contour label maps (H, W, T) -> filled masks, as the reference does it (cardiac_uda.py:223-246): per frame and
        class, the contour's pixels in np.argwhere (row-major) order are handed to cv2.fillPoly AS IF (row, col) were
        (x, y) -- so the polygon is filled in the transposed frame, its vertices in raster order rather than traced
        order -- and the filled pixels are transposed back.  Classes are the sorted distinct non-zero labels of the whole
        clip, at most ORGAN_NUM[view] of them; later classes overwrite earlier ones."""
        h, w, T = contours.shape
        all_cls = sorted(set(contours.reshape(-1).tolist()) - {0})
        out = np.zeros((h, w, T), dtype=np.float64)
        for i in range(T):
            contour = contours[:, :, i]
            for cls in range(1, ORGAN_NUM[self.view_num[0]] + 1):
                if cls > len(all_cls):
                    break
                pts = np.argwhere(contour == all_cls[cls - 1])
                if len(pts):
                    img = fill_poly(pts, (h, w))              # img[y][x] with x := row, y := col of the frame
                    xy = np.argwhere(img == 255)
                    out[xy[:, 1], xy[:, 0], i] = cls          # mask[idx[1], idx[0]] = cls: needs h == w like the reference
        return out

    def _clip(self, images, masks):
        """Strided clip of clip_length frames out of total_length (cardiac_uda.py:97-111)."""
        T = images.shape[-1]
        if T < self.clip_length:
            return None
        rate = int(self.total_length / self.clip_length)
        if T < self.clip_length * rate:
            rate = T // self.clip_length
        start = self.rng.randint(0, T - self.clip_length * rate)
        end = start + self.clip_length                    # sic: the reference's end is not scaled by the stride, so a
        sel = slice(start, end, rate)                     # stride > 1 yields fewer than clip_length frames
        m = masks[:, :, sel]
        index = np.where(np.sum(m, axis=(0, 1)) > 100, 1, 0)           # computed on the labels as stored (:108-110)
        if self.fill_mask:
            m = self.contour_to_mask(m)
        return images[:, :, sel], m, index

    def _load(self, index):
        entry = self.data_dict[self.id_list[(index // self.repeat) % max(self.num_data, 1)]]
        for k in self.view_num:
            img_p, msk_p = entry["images"].get(k), entry["masks"].get(k)
            if img_p is None or msk_p is None:
                continue
            images, masks = np.asarray(read_nifti(img_p)), np.asarray(read_nifti(msk_p))
            if self.single_frame:
                im, mk, mask_index = self.input_select(images, masks)
                if mask_index is None or np.sum(mk) < 100:                # cardiac_uda.py:83-86
                    continue
                return im, mk, mask_index
            if images.ndim == 3:
                got = self._clip(images, masks)
                if got is not None:
                    return got
        return None

    def __getitem__(self, index):
        if self.num_data == 0:
            raise IndexError("empty CardiacUDA selection")
        got, tries = self._load(index), 0
        while got is None:                                                # cardiac_uda.py:118-120
            tries += 1
            if tries > 8 * self.num_data:
                raise RuntimeError("no usable volume in the selection (no annotated frame / clip long enough)")
            index = self.rng.randint(0, self.num_data - 1) * self.repeat
            got = self._load(index)
        images, masks, mask_index = got
        labels = masks if self.seg_parts else np.where(masks > 0, 1, 0)   # :149-150
        return np.ascontiguousarray(images).astype(np.uint8)[None], np.ascontiguousarray(labels).astype(np.uint8), \
            mask_index, index
