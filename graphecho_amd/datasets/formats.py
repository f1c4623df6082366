"""MetaImage (.mhd/.raw/.zraw) and NIfTI-1 (.nii/.nii.gz) readers in plain numpy.

The reference reads these through SimpleITK (`skimage.io.imread(plugin='simpleitk')`, datasets/camus.py:110-112) and
nibabel (`np.array(nib.load(p).dataobj)`, datasets/cardiac_uda.py:80-81,92-94); neither library is in this image, so the
readers follow the published container layouts and return what those calls return:
  * read_mhd  -> array indexed [z][y][x] (DimSize reversed), the element type of the file;
  * read_nifti -> array of shape dim[1..ndim] (first index fastest on disk), scaled to float64 by scl_slope/scl_inter
    only when the header asks for it.
PARITY UNPINNED against SimpleITK/nibabel themselves (absent here); the tests pin byte-level vectors assembled from
the format specifications and write->read round trips.  The writers exist for those tests and for making fixtures.

EchoNet-Dynamic videos (datasets/echo.py:294-328, cv2.VideoCapture) are Motion-JPEG in an AVI (RIFF) container -- what
the reference's own savevideo writes (echo.py:344, fourcc 'MJPG').  read_avi_mjpeg walks the RIFF chunks and decodes
each '..dc' chunk as a baseline JPEG with Pillow (present in this image; cv2 / ffmpeg are not), bgr_to_gray restates
cv2.COLOR_BGR2GRAY's fixed-point formula.  Also unpinned: ffmpeg's and libjpeg's IDCT / chroma upsampling may differ
by a grey level.
"""
import gzip
import io
import os
import struct
import zlib

import numpy as np

_MET = {"MET_UCHAR": "u1", "MET_CHAR": "i1", "MET_USHORT": "u2", "MET_SHORT": "i2", "MET_UINT": "u4", "MET_INT": "i4",
        "MET_ULONG": "u4", "MET_LONG": "i4", "MET_ULONG_LONG": "u8", "MET_LONG_LONG": "i8", "MET_FLOAT": "f4",
        "MET_DOUBLE": "f8"}
_MET_OF = {"u1": "MET_UCHAR", "i1": "MET_CHAR", "u2": "MET_USHORT", "i2": "MET_SHORT", "u4": "MET_UINT",
           "i4": "MET_INT", "u8": "MET_ULONG_LONG", "i8": "MET_LONG_LONG", "f4": "MET_FLOAT", "f8": "MET_DOUBLE"}


def _truth(v):
    return v.strip().lower() in ("true", "1", "yes")


def read_mhd(path):
    """Parse a MetaImage header and its element data (raw or zlib-compressed, local or in a sibling file)."""
    with open(path, "rb") as f:
        blob = f.read()
    fields, pos = {}, 0
    while pos < len(blob):
        end = blob.find(b"\n", pos)
        end = len(blob) if end < 0 else end
        line = blob[pos:end].decode("latin-1").strip()
        pos = end + 1
        if "=" not in line:
            continue
        key, val = (s.strip() for s in line.split("=", 1))
        fields[key] = val
        if key == "ElementDataFile":          # by definition the last header line; LOCAL data starts right after it
            break
    for need in ("NDims", "DimSize", "ElementType", "ElementDataFile"):
        if need not in fields:
            raise ValueError(f"{path}: MetaImage header lacks {need}")
    if fields["ElementType"] not in _MET:
        raise ValueError(f"{path}: unsupported ElementType {fields['ElementType']}")
    dims = [int(v) for v in fields["DimSize"].split()][:int(fields["NDims"])]
    channels = int(fields.get("ElementNumberOfChannels", "1"))
    msb = _truth(fields.get("BinaryDataByteOrderMSB", fields.get("ElementByteOrderMSB", "False")))
    dtype = np.dtype((">" if msb else "<") + _MET[fields["ElementType"]])
    count = int(np.prod(dims)) * channels
    src = fields["ElementDataFile"]
    if src == "LOCAL":
        data = blob[pos:]
    else:
        if src.startswith("LIST") or "%" in src:
            raise ValueError(f"{path}: multi-file MetaImage element data is not supported")
        with open(os.path.join(os.path.dirname(path), src), "rb") as f:
            data = f.read()
        skip = int(fields.get("HeaderSize", "0"))
        if skip > 0:
            data = data[skip:]
        elif skip == -1 and not _truth(fields.get("CompressedData", "False")):
            data = data[len(data) - count * dtype.itemsize:]
    if _truth(fields.get("CompressedData", "False")):
        data = zlib.decompress(data)
    if len(data) < count * dtype.itemsize:
        raise ValueError(f"{path}: element data holds {len(data)} bytes, header needs {count * dtype.itemsize}")
    arr = np.frombuffer(data, dtype=dtype, count=count).astype(dtype.newbyteorder("="))
    shape = list(reversed(dims)) + ([channels] if channels > 1 else [])
    return arr.reshape(shape)


def write_mhd(path, arr, compressed=False, local=False, msb=False, spacing=None):
    """Write `arr` (indexed [z][y][x] or [y][x]) as a MetaImage header + .raw/.zraw sibling (or LOCAL data)."""
    arr = np.asarray(arr)
    code = arr.dtype.kind + str(arr.dtype.itemsize)
    if code not in _MET_OF:
        raise ValueError(f"unsupported dtype {arr.dtype}")
    dims = list(reversed(arr.shape))
    payload = arr.astype(arr.dtype.newbyteorder(">" if msb else "<")).tobytes()
    if compressed:
        payload = zlib.compress(payload)
    stem = os.path.splitext(os.path.basename(path))[0]
    data_name = "LOCAL" if local else stem + (".zraw" if compressed else ".raw")
    lines = ["ObjectType = Image", f"NDims = {len(dims)}", "BinaryData = True",
             f"BinaryDataByteOrderMSB = {msb}", f"CompressedData = {bool(compressed)}"]
    if compressed:
        lines.append(f"CompressedDataSize = {len(payload)}")
    lines += ["ElementSpacing = " + " ".join(str(s) for s in (spacing or [1] * len(dims))),
              "DimSize = " + " ".join(str(d) for d in dims), f"ElementType = {_MET_OF[code]}",
              f"ElementDataFile = {data_name}"]
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n").encode("latin-1"))
        if local:
            f.write(payload)
    if not local:
        with open(os.path.join(os.path.dirname(path), data_name), "wb") as f:
            f.write(payload)


_NIFTI_DT = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4", 1024: "i8", 1280: "u8"}
_NIFTI_OF = {v: k for k, v in _NIFTI_DT.items()}


def read_nifti(path):
    """Single-file NIfTI-1 (.nii, optionally gzip-compressed): the array `nibabel.load(path).dataobj` yields."""
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:2] == b"\x1f\x8b":
        blob = gzip.decompress(blob)
    if len(blob) < 348:
        raise ValueError(f"{path}: shorter than a NIfTI-1 header")
    if struct.unpack("<i", blob[:4])[0] == 348:
        en = "<"
    elif struct.unpack(">i", blob[:4])[0] == 348:
        en = ">"
    else:
        raise ValueError(f"{path}: sizeof_hdr is not 348 (not NIfTI-1)")
    if blob[344:347] not in (b"n+1", b"ni1"):
        raise ValueError(f"{path}: bad NIfTI magic {blob[344:348]!r}")
    if blob[344:347] == b"ni1":
        raise ValueError(f"{path}: two-file NIfTI (.hdr/.img) is not supported")
    dim = struct.unpack(en + "8h", blob[40:56])
    ndim = dim[0]
    if not 1 <= ndim <= 7:
        raise ValueError(f"{path}: dim[0] = {ndim}")
    shape = [int(d) for d in dim[1:1 + ndim]]
    datatype, bitpix = struct.unpack(en + "2h", blob[70:74])
    if datatype not in _NIFTI_DT:
        raise ValueError(f"{path}: unsupported NIfTI datatype {datatype}")
    vox_offset, slope, inter = struct.unpack(en + "3f", blob[108:120])
    dtype = np.dtype(en + _NIFTI_DT[datatype])
    if dtype.itemsize * 8 != bitpix:
        raise ValueError(f"{path}: bitpix {bitpix} does not match datatype {datatype}")
    start = max(int(vox_offset), 352)
    count = int(np.prod(shape))
    if len(blob) < start + count * dtype.itemsize:
        raise ValueError(f"{path}: voxel data truncated")
    arr = np.frombuffer(blob, dtype=dtype, count=count, offset=start).astype(dtype.newbyteorder("="))
    arr = arr.reshape(shape, order="F")
    if np.isfinite(slope) and slope != 0.0 and not (slope == 1.0 and inter == 0.0):
        arr = arr.astype(np.float64) * float(slope) + float(inter)
    return arr


def write_nifti(path, arr, slope=0.0, inter=0.0, big_endian=False):
    """Write `arr` as a single-file NIfTI-1 (gzip when the name ends in .gz)."""
    arr = np.asarray(arr)
    code = arr.dtype.kind + str(arr.dtype.itemsize)
    if code not in _NIFTI_OF:
        raise ValueError(f"unsupported dtype {arr.dtype}")
    en = ">" if big_endian else "<"
    hdr = bytearray(348)
    struct.pack_into(en + "i", hdr, 0, 348)
    dim = [arr.ndim] + list(arr.shape) + [1] * (7 - arr.ndim)
    struct.pack_into(en + "8h", hdr, 40, *dim)
    struct.pack_into(en + "2h", hdr, 70, _NIFTI_OF[code], arr.dtype.itemsize * 8)
    struct.pack_into(en + "8f", hdr, 76, 1.0, *([1.0] * 7))
    struct.pack_into(en + "3f", hdr, 108, 352.0, slope, inter)
    hdr[344:348] = b"n+1\x00"
    blob = bytes(hdr) + b"\x00" * 4 + arr.astype(arr.dtype.newbyteorder(en)).tobytes(order="F")
    if path.endswith(".gz"):
        blob = gzip.compress(blob, compresslevel=1)
    with open(path, "wb") as f:
        f.write(blob)


# ---------------------------------------------------------------------------------------------------------------------
# AVI (RIFF) + Motion-JPEG
# ---------------------------------------------------------------------------------------------------------------------
def _riff_chunks(blob, start, end):
    """Yield (fourcc, list type or None, data start, data end) for the chunks in blob[start:end]."""
    pos = start
    while pos + 8 <= end:
        cid = blob[pos:pos + 4]
        size = struct.unpack_from("<I", blob, pos + 4)[0]
        d0, d1 = pos + 8, min(pos + 8 + size, end)
        if cid in (b"RIFF", b"LIST"):
            yield cid, blob[d0:d0 + 4], d0 + 4, d1
        else:
            yield cid, None, d0, d1
        pos = d0 + size + (size & 1)             # chunks are word aligned


_STD_DHT = None


def _standard_dht():
    """The JPEG Annex K Huffman tables as DHT segments (Motion-JPEG streams may omit them): taken from a JPEG Pillow
    writes with optimize=False, which uses exactly those tables."""
    global _STD_DHT
    if _STD_DHT is None:
        from PIL import Image

        buf = io.BytesIO()
        Image.new("RGB", (16, 16), (120, 60, 200)).save(buf, "JPEG", quality=75, optimize=False)
        raw, pos, out = buf.getvalue(), 2, b""
        while pos + 4 <= len(raw) and raw[pos] == 0xFF and raw[pos + 1] != 0xDA:
            seg = struct.unpack_from(">H", raw, pos + 2)[0]
            if raw[pos + 1] == 0xC4:
                out += raw[pos:pos + 2 + seg]
            pos += 2 + seg
        _STD_DHT = out
    return _STD_DHT


def _decode_jpeg(data):
    from PIL import Image

    if b"\xff\xc4" not in data[:data.find(b"\xff\xda") if b"\xff\xda" in data else len(data)]:
        sos = data.find(b"\xff\xda")            # no DHT before the scan: splice the standard tables in
        if sos > 0:
            data = data[:sos] + _standard_dht() + data[sos:]
    img = Image.open(io.BytesIO(data))
    return np.asarray(img.convert("RGB"))


def read_avi_mjpeg(path):
    """-> uint8 (frames, height, width, 3) RGB.  Raises ValueError for anything but Motion-JPEG video streams."""
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:4] != b"RIFF" or blob[8:12] != b"AVI ":
        raise ValueError(f"{path}: not a RIFF/AVI file")
    handler, frames = None, []
    stack = [(12, len(blob))]
    while stack:
        a, b = stack.pop()
        for cid, ltype, d0, d1 in _riff_chunks(blob, a, b):
            if ltype is not None:
                if ltype == b"movi":
                    for fid, ftype, f0, f1 in _riff_chunks(blob, d0, d1):
                        if ftype is not None:              # 'rec ' lists group the frame chunks of some writers
                            frames += [(g0, g1) for gid, gt, g0, g1 in _riff_chunks(blob, f0, f1)
                                       if gt is None and gid[2:] in (b"dc", b"db")]
                        elif fid[2:] in (b"dc", b"db"):
                            frames.append((f0, f1))
                else:
                    stack.append((d0, d1))
            elif cid == b"strh" and blob[d0:d0 + 4] == b"vids":
                handler = blob[d0 + 4:d0 + 8]
    if handler is not None and handler.upper() not in (b"MJPG", b"JPEG", b"\x00\x00\x00\x00"):
        raise ValueError(f"{path}: video stream is {handler!r}, only Motion-JPEG ('MJPG') can be decoded without a codec library")
    out, last = [], None
    for f0, f1 in frames:
        if f1 > f0:
            last = _decode_jpeg(blob[f0:f1])
        if last is None:
            raise ValueError(f"{path}: first frame is empty")
        out.append(last)                                  # a zero-length chunk repeats the previous frame
    if not out:
        raise ValueError(f"{path}: no video frames")
    return np.stack(out)


def write_avi_mjpeg(path, frames, fps=50, quality=95):
    """frames: uint8 (F, H, W, 3) RGB -> Motion-JPEG AVI (the layout cv2.VideoWriter('MJPG') produces: hdrl with avih /
    strh / strf, movi with one '00dc' chunk per frame, idx1)."""
    from PIL import Image

    frames = np.asarray(frames, dtype=np.uint8)
    F, H, W, _ = frames.shape
    jpegs = []
    for fr in frames:
        buf = io.BytesIO()
        Image.fromarray(fr, "RGB").save(buf, "JPEG", quality=quality, optimize=False)
        jpegs.append(buf.getvalue())
    chunk = lambda cid, data: cid + struct.pack("<I", len(data)) + data + (b"\x00" if len(data) & 1 else b"")
    lst = lambda t, data: chunk(b"LIST", t + data)
    maxb = max(len(j) for j in jpegs)
    avih = struct.pack("<14I", int(1e6 / fps), maxb * int(fps), 0, 0x10, F, 0, 1, maxb, W, H, 0, 0, 0, 0)
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, 1, int(fps), 0, F, maxb, 0xFFFFFFFF, 0, 0, 0, W, H)
    strf = struct.pack("<IiiHH4sIiiII", 40, W, H, 1, 24, b"MJPG", W * H * 3, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi_body, idx, off = b"", b"", 4
    for j in jpegs:
        c = chunk(b"00dc", j)
        idx += b"00dc" + struct.pack("<III", 0x10, off, len(j))
        movi_body += c
        off += len(c)
    body = b"AVI " + hdrl + lst(b"movi", movi_body) + chunk(b"idx1", idx)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def bgr_to_gray(rgb):
    """cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY) on an (…, 3) uint8 array given here in R, G, B order: OpenCV's 14-bit
    fixed point  Y = (R * 4899 + G * 9617 + B * 1868 + 8192) >> 14  (0.299 / 0.587 / 0.114)."""
    a = np.asarray(rgb).astype(np.int32)
    return ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)
