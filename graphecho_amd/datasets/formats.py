"""MetaImage (.mhd/.raw/.zraw) and NIfTI-1 (.nii/.nii.gz) readers in plain numpy.

The reference reads these through SimpleITK (`skimage.io.imread(plugin='simpleitk')`, datasets/camus.py:110-112) and
nibabel (`np.array(nib.load(p).dataobj)`, datasets/cardiac_uda.py:80-81,92-94); neither library is in this image, so the
readers follow the published container layouts and return what those calls return:
  * read_mhd  -> array indexed [z][y][x] (DimSize reversed), the element type of the file;
  * read_nifti -> array of shape dim[1..ndim] (first index fastest on disk), scaled to float64 by scl_slope/scl_inter
    only when the header asks for it.
PARITY UNPINNED against SimpleITK/nibabel themselves (absent here); the tests pin byte-level vectors assembled from
the format specifications and write->read round trips.  The writers exist for those tests and for making fixtures.
"""
import gzip
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
