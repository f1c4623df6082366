"""On-disk decoders: MetaImage / NIfTI-1 readers against byte vectors assembled from the format
specifications and write->read round trips; CAMUS / CardiacUDA sample logic on temporary trees (reference
datasets/camus.py, datasets/cardiac_uda.py).  SimpleITK / nibabel are absent here: parity with them is unpinned."""
import gzip
import os
import random
import struct
import zlib

import numpy as np
import pytest

from graphecho_amd.datasets import CamusSet, CardiacUDASet, read_mhd, read_nifti, write_mhd, write_nifti


def test_mhd_byte_vectors(tmp_path):
    # 3 columns x 2 rows, uint8, separate raw file: DimSize lists x first, the array is indexed [y][x]
    (tmp_path / "a.mhd").write_text("ObjectType = Image\nNDims = 2\nDimSize = 3 2\nElementType = MET_UCHAR\n"
                                    "ElementDataFile = a.raw\n")
    (tmp_path / "a.raw").write_bytes(bytes([1, 2, 3, 4, 5, 6]))
    assert read_mhd(str(tmp_path / "a.mhd")).tolist() == [[1, 2, 3], [4, 5, 6]]
    # big-endian int16, LOCAL data after the header, a trailing singleton z like CAMUS stills (DimSize = W H 1)
    hdr = ("ObjectType = Image\nNDims = 3\nDimSize = 2 2 1\nElementType = MET_SHORT\n"
           "BinaryDataByteOrderMSB = True\nElementDataFile = LOCAL\n").encode()
    (tmp_path / "b.mhd").write_bytes(hdr + struct.pack(">4h", -2, 300, 7, 1000))
    b = read_mhd(str(tmp_path / "b.mhd"))
    assert b.shape == (1, 2, 2) and b.dtype == np.int16 and b[0].tolist() == [[-2, 300], [7, 1000]]
    # zlib-compressed element file
    (tmp_path / "c.mhd").write_text("NDims = 2\nDimSize = 2 2\nElementType = MET_UCHAR\nCompressedData = True\n"
                                    "ElementDataFile = c.zraw\n")
    (tmp_path / "c.zraw").write_bytes(zlib.compress(bytes([9, 8, 7, 6])))
    assert read_mhd(str(tmp_path / "c.mhd")).tolist() == [[9, 8], [7, 6]]
    (tmp_path / "bad.mhd").write_text("NDims = 2\nDimSize = 2 2\nElementType = MET_UCHAR\nElementDataFile = c.zraw\n")
    (tmp_path / "short.raw").write_bytes(b"\x00")
    (tmp_path / "short.mhd").write_text("NDims = 2\nDimSize = 2 2\nElementType = MET_UCHAR\nElementDataFile = short.raw\n")
    with pytest.raises(ValueError):
        read_mhd(str(tmp_path / "short.mhd"))


@pytest.mark.parametrize("dtype", ["u1", "i2", "f4"])
@pytest.mark.parametrize("mode", ["raw", "zraw", "local", "msb"])
def test_mhd_round_trip(tmp_path, dtype, mode):
    rng = np.random.default_rng(1)
    arr = (rng.integers(0, 200, (3, 5, 7)) if dtype != "f4" else rng.standard_normal((3, 5, 7))).astype(dtype)
    p = str(tmp_path / "v.mhd")
    write_mhd(p, arr, compressed=mode == "zraw", local=mode == "local", msb=mode == "msb")
    got = read_mhd(p)
    assert got.dtype == arr.dtype and np.array_equal(got, arr)


def _nifti_blob(shape, dtype_code, bitpix, payload, en="<", slope=0.0, inter=0.0, vox_offset=352.0):
    """NIfTI-1 header by field offset (sizeof_hdr 0, dim 40, datatype 70, bitpix 72, vox_offset 108, scl_slope 112,
    scl_inter 116, magic 344) -- written independently of write_nifti."""
    h = bytearray(348)
    struct.pack_into(en + "i", h, 0, 348)
    struct.pack_into(en + "8h", h, 40, len(shape), *(list(shape) + [1] * (7 - len(shape))))
    struct.pack_into(en + "h", h, 70, dtype_code)
    struct.pack_into(en + "h", h, 72, bitpix)
    struct.pack_into(en + "f", h, 108, vox_offset)
    struct.pack_into(en + "f", h, 112, slope)
    struct.pack_into(en + "f", h, 116, inter)
    h[344:348] = b"n+1\x00"
    return bytes(h) + b"\x00" * (int(vox_offset) - 348) + payload


def test_nifti_byte_vectors(tmp_path):
    # uint8 2 x 3 volume: the FIRST index is the fastest on disk
    p = tmp_path / "a.nii"
    p.write_bytes(_nifti_blob((2, 3), 2, 8, bytes([0, 1, 2, 3, 4, 5])))
    a = read_nifti(str(p))
    assert a.dtype == np.uint8 and a.shape == (2, 3) and a.tolist() == [[0, 2, 4], [1, 3, 5]]
    # big-endian int16 with intensity scaling -> float64 = raw * slope + inter; gzip container
    p = tmp_path / "b.nii.gz"
    p.write_bytes(gzip.compress(_nifti_blob((2, 1, 2), 4, 16, struct.pack(">4h", 1, -2, 3, 4), en=">", slope=0.5, inter=10.0)))
    b = read_nifti(str(p))
    assert b.dtype == np.float64 and b.shape == (2, 1, 2) and b[:, 0, :].tolist() == [[10.5, 11.5], [9.0, 12.0]]
    # slope 0 means "no scaling"; data may start beyond 352
    p = tmp_path / "c.nii"
    p.write_bytes(_nifti_blob((4,), 16, 32, struct.pack("<4f", 1.5, 2.5, -1.0, 0.0), vox_offset=400.0))
    assert read_nifti(str(p)).tolist() == [1.5, 2.5, -1.0, 0.0]
    (tmp_path / "bad.nii").write_bytes(b"\x00" * 400)
    with pytest.raises(ValueError):
        read_nifti(str(tmp_path / "bad.nii"))


@pytest.mark.parametrize("name,kw", [("v.nii", {}), ("v.nii.gz", {}), ("w.nii", {"big_endian": True}),
                                     ("s.nii.gz", {"slope": 2.0, "inter": -1.0})])
def test_nifti_round_trip(tmp_path, name, kw):
    arr = np.random.default_rng(2).integers(0, 255, (6, 5, 4)).astype(np.uint8)
    p = str(tmp_path / name)
    write_nifti(p, arr, **kw)
    got = read_nifti(p)
    if "slope" in kw:
        assert got.dtype == np.float64 and np.array_equal(got, arr * 2.0 - 1.0)
    else:
        assert got.dtype == np.uint8 and np.array_equal(got, arr)


def _camus_tree(root, n=12, hw=(30, 40)):
    rng = np.random.default_rng(0)
    for i in range(n):
        pid = f"patient{i:04d}"
        d = root / "training" / pid
        d.mkdir(parents=True)
        if i == 5:
            continue                                   # empty directory: skipped by the listing (camus.py:55-56)
        for view in ("4CH_ED", "4CH_ED_gt"):
            if i == 7 and view == "4CH_ED":
                continue                               # missing frame: __getitem__ redraws another patient
            h, w = hw[0] + i, hw[1] + 2 * i
            arr = (rng.integers(0, 4, (1, h, w)) if view.endswith("gt") else rng.integers(0, 256, (1, h, w))).astype(np.uint8)
            write_mhd(str(d / f"{pid}_{view}.mhd"), arr, compressed=(i % 2 == 0))
    return root


def test_camus_split_and_samples(tmp_path):
    root = str(_camus_tree(tmp_path))
    sets = {s: CamusSet(root, "4CH_ED", "4CH_ED_gt", s) for s in ("train", "valid", "test")}
    listed = sorted(os.listdir(os.path.join(root, "training")))
    patients = [os.path.join(root, "training", p) for p in listed if p != "patient0005"]
    random.Random(123).shuffle(patients)               # camus.py:60-67 with train_ratio 1.0, valid_ratio 0.2
    assert sets["train"].data_list == patients[2:11] and sets["valid"].data_list == patients[:1]
    assert sets["test"].data_list == patients[1:2]
    assert CamusSet(root, "4CH_ED", "4CH_ED_gt", "train").data_list == sets["train"].data_list     # deterministic
    tr = sets["train"]
    assert tr.class_values == (1, 3)
    for i in range(len(tr)):
        frame, label, mask_index, idx = tr[i]
        pid = os.path.basename(tr.data_list[idx])
        assert "0007" not in pid                      # the patient without a frame is never returned
        k = int(pid[-4:])
        assert frame.shape == (1, 30 + k, 40 + 2 * k) and label.shape == frame.shape[1:] and frame.dtype == np.uint8
        assert np.array_equal(label, read_mhd(os.path.join(tr.data_list[idx], f"{pid}_4CH_ED_gt.mhd"))[0])
        assert mask_index == 0
    with pytest.raises(ValueError):
        CamusSet(root, "4CH_ED", "4CH_ED_gt", "nope")


def _uda_tree(root, n=10, T=20):
    rng = np.random.default_rng(3)
    infos = {}
    for i in range(n):
        H, W = 40 + i, 50 + i
        img = rng.integers(0, 256, (H, W, T)).astype(np.uint8)
        lab = np.zeros((H, W, T), np.uint8)
        for t in (3, 4, 11):                            # annotated frames: > 100 labelled pixels
            lab[5:25, 5:25, t] = 1
            lab[26:36, 10:30, t] = 2
        lab[0, 0, 7] = 1                                # a stray pixel: frame 7 is NOT "annotated"
        pi, pm = str(root / f"img{i}.nii.gz"), str(root / f"lab{i}.nii.gz")
        write_nifti(pi, img)
        write_nifti(pm, lab)
        infos[f"id{i}"] = {"dataset_name": "Site_G" if i < 8 else "Site_R",
                           "views_images": {"4": pi, "2": None}, "views_labels": {"4": pm, "2": None}}
    return infos


def test_cardiac_uda_single_frames_and_clips(tmp_path):
    infos = _uda_tree(tmp_path)
    ds = CardiacUDASet(infos, str(tmp_path), True, set_select=("Site_G",), view_num=("4",))
    assert ds.class_values == (0, 1, 2, 3, 4) and ds.num_data == 7 and len(ds.valid_list) == 0 and len(ds.test_list) == 1
    assert set(ds.id_list) <= {f"id{i}" for i in range(8)}
    seen = set()
    for i in range(len(ds)):
        frame, label, t, idx = ds[i]
        k = int(ds.id_list[idx][2:])
        assert frame.shape == (1, 40 + k, 50 + k) and label.shape == (40 + k, 50 + k)
        assert t in (3, 4, 11) and label.sum() > 100 and set(np.unique(label)) == {0, 1, 2}
        assert np.array_equal(frame[0], read_nifti(infos[f"id{k}"]["views_images"]["4"])[:, :, t])
        seen.add(t)
    # same seed -> same draws; evaluation split takes an explicit id list
    again = CardiacUDASet(infos, str(tmp_path), True, set_select=("Site_G",), view_num=("4",))
    assert again.id_list == ds.id_list and again[0][2] == CardiacUDASet(infos, str(tmp_path), True, view_num=("4",))[0][2]
    ev = CardiacUDASet(infos, str(tmp_path), False, data_list=ds.test_list, view_num=("4",))
    assert len(ev) == 1 and ev[0][0].ndim == 3
    # a view nobody annotated: every draw fails and the error says so
    with pytest.raises(RuntimeError):
        CardiacUDASet(infos, str(tmp_path), True, view_num=("2",))[0]
    # clips: clip_length 8 out of total_length 16 -> stride 2, and the reference's unscaled end index gives 4 frames
    clips = CardiacUDASet(infos, str(tmp_path), True, view_num=("4",), single_frame=False, total_length=16, clip_length=8)
    frames, labels, mask_index, _ = clips[0]
    assert frames.shape[-1] == 4 and labels.shape == frames.shape[1:] and mask_index.shape == (4,)
    clips1 = CardiacUDASet(infos, str(tmp_path), True, view_num=("4",), single_frame=False, total_length=8, clip_length=8)
    frames, labels, mask_index, _ = clips1[0]
    assert frames.shape[-1] == 8 and frames.dtype == np.uint8
    binary = CardiacUDASet(infos, str(tmp_path), True, view_num=("4",), seg_parts=False)
    assert binary.class_values is None and set(np.unique(binary[0][1])) == {0, 1}
    filled = CardiacUDASet(infos, str(tmp_path), True, view_num=("4",), single_frame=False, total_length=8,
                           clip_length=8, fill_mask=True)
    assert filled.fill_mask and filled[0][1].shape == clips1[0][1].shape


def test_raw_batches_shards_are_disjoint_and_equal_length():
    """Data-parallel sharding of the batch iterator: same permutation on every rank, disjoint samples, equal counts."""
    import torch
    from graphecho_amd.datasets import RawBatches

    class Toy:
        def __len__(self):
            return 23

        def __getitem__(self, i):
            return np.full((1, 2, 2), i, np.uint8), np.full((2, 2), i, np.uint8), 0, i

    seen = []
    for rank in range(3):
        it = RawBatches(Toy(), 2, torch.device("cpu"), shuffle=True, drop_last=True, seed=4, rank=rank, world=3)
        ids = [int(f[0, 0, 0, 0]) for frames, _ in it for f in frames]
        assert len(it) == 3 and len(ids) == 6          # 23 // 3 = 7 samples per rank -> 3 full batches of 2
        seen.append(set(ids))
    assert not (seen[0] & seen[1]) and not (seen[0] & seen[2]) and not (seen[1] & seen[2])
    single = RawBatches(Toy(), 4, torch.device("cpu"))
    assert len(single) == 6 and sum(len(f) for f, _ in single) == 23


# ---- rasterisers, EchoNet-Dynamic ------------------------------------------------------------------------------
def test_polygon_matches_crossing_number_definition():
    """datasets/raster.py:polygon (skimage.draw.polygon of scikit-image 0.18 restated): a pixel is inside iff a ray to
    +x crosses the boundary an odd number of times, with the library's half-open edge rule -- checked against an
    independent per-pixel evaluation, on a convex, a concave and a clipped polygon; and the known answers of an
    axis-aligned square (top/left edges in, bottom/right edges out)."""
    from graphecho_amd.datasets import polygon

    def brute(r, c, shape):
        out = np.zeros(shape, bool)
        n = len(r)
        for y in range(shape[0]):
            for x in range(shape[1]):
                inside, j = False, n - 1
                for i in range(n):
                    if ((r[i] <= y < r[j]) or (r[j] <= y < r[i])) and x < (c[j] - c[i]) * (y - r[i]) / (r[j] - r[i]) + c[i]:
                        inside = not inside
                    j = i
                out[y, x] = inside
        return out

    for r, c, shape in [((2, 2, 9, 9), (3, 12, 12, 3), (14, 16)), ((1, 8, 15, 8, 8), (8, 14, 8, 2, 7), (17, 17)),
                        ((-3, 5, 30, 11), (4, 25, 9, -6), (20, 18))]:
        rr, cc = polygon(r, c, shape)
        got = np.zeros(shape, bool)
        got[rr, cc] = True
        assert np.array_equal(got, brute(r, c, shape))
        assert np.array_equal(np.lexsort((cc, rr)), np.arange(len(rr)))          # row-major order
    rr, cc = polygon((2, 2, 9, 9), (3, 12, 12, 3), (14, 16))
    assert (rr.min(), rr.max(), cc.min(), cc.max()) == (2, 8, 3, 11)             # [2, 9) x [3, 12)
    assert len(rr) == 7 * 9


def test_fill_poly_matches_opencv_properties():
    """datasets/raster.py:fill_poly (cv2.fillPoly restated): boundary edges drawn as 8-connected lines, interior filled by
    the even-odd rule -- for convex polygons with integer vertices that is exactly the closed polygon's lattice points;
    a self-intersecting (bow-tie) vertex order leaves the doubly covered... lobes filled and the crossing point set."""
    from graphecho_amd.datasets import fill_poly

    img = fill_poly([(3, 2), (12, 2), (12, 9), (3, 9)], (14, 16))
    ref = np.zeros((14, 16), np.uint8)
    ref[2:10, 3:13] = 255                                                          # closed rectangle: all four edges in
    assert np.array_equal(img, ref)
    tri = fill_poly([(1, 1), (13, 1), (1, 13)], (16, 16))
    yy, xx = np.mgrid[0:16, 0:16]
    assert np.array_equal(tri == 255, (xx >= 1) & (yy >= 1) & (xx + yy <= 14))     # lattice points of the closed triangle
    bow = fill_poly([(2, 2), (12, 12), (12, 2), (2, 12)], (15, 15))
    assert bow[7, 7] == 255 and bow[7, 3] == 255 and bow[7, 11] == 255 and bow[3, 7] == 0 and bow[11, 7] == 0
    one = fill_poly([(5, 6)], (10, 10))
    assert one.sum() == 255 and one[6, 5] == 255                                   # a single vertex is its own outline
    clipped = fill_poly([(-5, -5), (30, -5), (30, 30), (-5, 30)], (8, 9))
    assert (clipped == 255).all()


def test_contour_to_mask_fills_in_the_transposed_frame(tmp_path):
    """CardiacUDA fill_mask (cardiac_uda.py:223-246): contour pixels in argwhere order through fillPoly with (row, col)
    taken as (x, y).  For a closed rectangle outline the raster-order polygon still covers the rectangle: every pixel of
    the outline stays set and the inside is filled wherever the even-odd walk over the raster-ordered vertices covers
    it; the result is transposed back (a non-square outline lands on its own rows/columns, not the transpose)."""
    from graphecho_amd.datasets import CardiacUDASet

    ds = CardiacUDASet({}, str(tmp_path), False, data_list=[], view_num=("2",), single_frame=False, fill_mask=True)
    c = np.zeros((24, 24, 2), np.uint8)
    c[4, 6:17, 0] = c[12, 6:17, 0] = 1                # outline of rows 4..12 x cols 6..16
    c[4:13, 6, 0] = c[4:13, 16, 0] = 1
    m = ds.contour_to_mask(c)
    assert m.shape == (24, 24, 2) and m[:, :, 1].sum() == 0
    assert (m[:, :, 0][c[:, :, 0] == 1] == 1).all(), "outline pixels are vertices: always set"
    assert m[:, :, 0][:4].sum() == 0 and m[:, :, 0][13:].sum() == 0 and m[:, :, 0][:, :6].sum() == 0 and m[:, :, 0][:, 17:].sum() == 0
    assert m[8, 11, 0] == 1                           # interior reached by the raster-order polygon


def _write_echonet(root, rng):
    from graphecho_amd.datasets import write_avi_mjpeg

    os.makedirs(os.path.join(root, "Videos"))
    rows, traces = [], []
    for vid, split, nfr, traced in (("0XAAA", "TRAIN", 12, (3, 7)), ("0XBBB", "TRAIN", 9, (6, 2)), ("0XCCC", "VAL", 10, (1, 4)),
                                    ("0XDDD", "TRAIN", 10, (5,))):
        yy, xx = np.mgrid[0:112, 0:112]
        frames = np.stack([np.clip(60 + 40 * np.sin((xx + 3 * t) / 9.0) + 30 * np.cos(yy / 7.0) + rng.normal(0, 2, (112, 112)), 0, 255)
                           for t in range(nfr)]).astype(np.uint8)
        write_avi_mjpeg(os.path.join(root, "Videos", vid + ".avi"), np.repeat(frames[..., None], 3, axis=-1), fps=50)
        rows.append(f"{vid},55.0,30.0,70.0,112,112,50,{nfr},{split}")
        for fr in traced:
            for k in range(21):                        # 21 chords: the first is the long axis, the rest span the cavity
                y = 20 + 3.5 * k
                half = 6 + 18 * np.sin(np.pi * (k + 1) / 22)
                traces.append(f"{vid}.avi,{56 - half + fr:.3f},{y:.3f},{56 + half + fr:.3f},{y:.3f},{fr}")
    with open(os.path.join(root, "FileList.csv"), "w") as f:
        f.write("FileName,EF,ESV,EDV,FrameHeight,FrameWidth,FPS,NumberOfFrames,Split\n" + "\n".join(rows) + "\n")
    with open(os.path.join(root, "VolumeTracings.csv"), "w") as f:
        f.write("FileName,X1,Y1,X2,Y2,Frame\n" + "\n".join(traces) + "\n")


def test_avi_mjpeg_round_trip_and_grey(tmp_path):
    """RIFF/AVI walk + JPEG decode: frame count, size and content (JPEG quality 95: within a few grey levels) survive
    write -> read; a stream without Huffman tables (the 'AVI1' flavour of Motion-JPEG) decodes after the standard
    tables are spliced in; cv2's BGR->grey fixed-point formula on known triples."""
    from graphecho_amd.datasets import bgr_to_gray, read_avi_mjpeg, write_avi_mjpeg
    from graphecho_amd.datasets import formats

    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:48, 0:64]
    frames = np.stack([np.stack([(xx * 3 + 10 * t) % 256, (yy * 5) % 256, (xx + yy) % 256], -1) for t in range(5)]).astype(np.uint8)
    frames = (frames // 8 * 8).astype(np.uint8)
    smooth = np.stack([np.repeat((100 + 50 * np.sin((xx + 4 * t) / 11.0))[..., None], 3, -1) for t in range(5)]).astype(np.uint8)
    p = str(tmp_path / "v.avi")
    write_avi_mjpeg(p, smooth)
    back = read_avi_mjpeg(p)
    assert back.shape == smooth.shape and back.dtype == np.uint8
    assert np.abs(back.astype(int) - smooth.astype(int)).max() <= 4
    raw = open(p, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"AVI " and b"MJPG" in raw[:300] and raw.count(b"00dc") >= 5
    # a frame without its DHT segments (Motion-JPEG streams may omit them) decodes after the standard tables are spliced in
    j0 = raw.find(b"\xff\xd8")
    j1 = raw.find(b"\xff\xd9", j0) + 2
    jpeg = raw[j0:j1]
    k, bare = 2, jpeg[:2]
    while jpeg[k + 1] != 0xDA:
        seg = int.from_bytes(jpeg[k + 2:k + 4], "big")
        if jpeg[k + 1] != 0xC4:
            bare += jpeg[k:k + 2 + seg]
        k += 2 + seg
    bare += jpeg[k:]
    assert b"\xff\xc4" not in bare[:bare.find(b"\xff\xda")]
    assert np.array_equal(formats._decode_jpeg(bare), formats._decode_jpeg(jpeg))
    g = bgr_to_gray(np.array([[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 90]], np.uint8))
    assert g.tolist() == [255, 0, 76, 150, 29, 131]                # 0.299 R + 0.587 G + 0.114 B, cv2's rounding
    with pytest.raises(ValueError):
        bad = str(tmp_path / "x.avi")
        open(bad, "wb").write(raw.replace(b"vidsMJPG", b"vidsH264"))
        read_avi_mjpeg(bad)


def test_echonet_set(tmp_path):
    """EchoSet against a synthetic EchoNet-Dynamic tree: split filter, videos with fewer than two traced frames dropped,
    clip = 8 frames starting AT the traced frame (LargeTrace: last traced, SmallTrace: first), zero padding past the end,
    LV mask = the tracing polygon (chords 1.., left ends down, right ends up), sample layout of the clip-fold kernel."""
    from graphecho_amd.datasets import EchoSet, RawBatches, read_avi_mjpeg, bgr_to_gray

    root = str(tmp_path / "echonet")
    _write_echonet(root, np.random.default_rng(1))
    tr = EchoSet(root, "train")
    assert [os.path.splitext(n)[0] for n in tr.fnames] == ["0XAAA", "0XBBB"]       # 0XDDD has one traced frame: dropped
    assert len(EchoSet(root, "val")) == 1 and len(EchoSet(root, "all")) == 3
    clip, mask, _, idx = tr[0]
    assert clip.shape == (1, 112, 112, 8) and clip.dtype == np.uint8 and mask.shape == (112, 112) and idx == 0
    video = bgr_to_gray(read_avi_mjpeg(os.path.join(root, "Videos", "0XAAA.avi")))
    assert np.array_equal(clip[0, :, :, 0], video[7]) and np.array_equal(clip[0, :, :, 4], video[11])
    assert clip[0, :, :, 5:].max() == 0                                             # frames 12.. do not exist: black
    small = EchoSet(root, "train", target_type="SmallTrace")[0]
    assert np.array_equal(small[0][0, :, :, 0], video[3])
    # mask: rows 24..90 (chords 1..20, y = 23.5 .. 90), widest around the middle, shifted by the frame number (+7)
    rows = np.nonzero(mask.any(1))[0]
    assert 22 <= rows.min() <= 25 and 88 <= rows.max() <= 90
    cols = np.nonzero(mask[56])[0]
    assert abs((cols.min() + cols.max()) / 2 - (56 + 7)) <= 1 and 40 <= cols.max() - cols.min() <= 50
    assert mask[10].sum() == 0 and mask[100].sum() == 0 and set(np.unique(mask)) == {0, 1}
    frames, labels = next(iter(RawBatches(tr, 2, __import__("torch").device("cpu"))))
    assert frames[0].shape == (1, 1, 112, 112, 8) and labels[0].shape == (1, 112, 112)
    with pytest.raises(FileNotFoundError):
        os.remove(os.path.join(root, "Videos", "0XBBB.avi"))
        EchoSet(root, "train")
