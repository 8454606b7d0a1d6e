"""Minimal NIfTI-1 single-file (.nii / .nii.gz) writer and reader.

The reference exports its parameter maps through nibabel (`nib.Nifti1Image(images, None)` +
`nib.save(..., name + '.nii.gz')`, model.py:790-801).  nibabel is not a dependency here; the format
is a fixed 348-byte header (NIfTI-1.1 specification, nifti1.h) + 4 bytes of extension flags + the
voxel array in Fortran (x fastest) order, optionally gzipped.  Only what save_predictions needs:
float32 / float64 / int16 / uint8 arrays of 1-7 dimensions, little-endian, unit voxel sizes and no
orientation (qform_code = sform_code = 0) unless a template header is supplied -- what
`Nifti1Image(arr, None)` writes.
"""
import gzip
import struct

import numpy as np

_DTYPES = {np.dtype(np.uint8): (2, 8), np.dtype(np.int16): (4, 16), np.dtype(np.int32): (8, 32),
           np.dtype(np.float32): (16, 32), np.dtype(np.float64): (64, 64)}
_CODES = {code: dt for dt, (code, _) in _DTYPES.items()}
HEADER_BYTES = 348
VOX_OFFSET = 352


def make_header(shape, dtype, template=None):
    """348-byte NIfTI-1 header for an array of `shape`/`dtype`.  With `template` (the bytes of
    another header, e.g. transform_directory/example.nii.gz as in model.py:794-797) orientation,
    voxel sizes and units are inherited and only dim / datatype / bitpix are replaced."""
    dtype = np.dtype(dtype)
    if dtype not in _DTYPES:
        raise ValueError(f"unsupported NIfTI datatype {dtype}")
    if not 1 <= len(shape) <= 7:
        raise ValueError("NIfTI-1 stores 1 to 7 dimensions")
    code, bitpix = _DTYPES[dtype]
    dim = [len(shape)] + list(shape) + [1] * (7 - len(shape))
    if template is not None:
        if len(template) < HEADER_BYTES or struct.unpack_from("<i", template, 0)[0] != HEADER_BYTES:
            raise ValueError("template is not a little-endian NIfTI-1 header")
        h = bytearray(template[:HEADER_BYTES])
        # only the geometry is inherited (pixdim, qform / sform, xyzt_units): the template's intensity scaling
        # and display window describe ITS voxel values (e.g. a scaled int16 scan) and would rescale the float
        # maps written under it in every reader -- nibabel's Nifti1Image(arr, None, header=hdr) resets them too
        struct.pack_into("<f", h, 112, 1.0)                               # scl_slope
        struct.pack_into("<f", h, 116, 0.0)                               # scl_inter
        struct.pack_into("<ff", h, 124, 0.0, 0.0)                         # cal_max, cal_min
        struct.pack_into("<h", h, 68, 0)                                  # intent_code
        struct.pack_into("<3f", h, 56, 0.0, 0.0, 0.0)                     # intent_p1..p3
        h[148:228] = b"\0" * 80                                           # descrip
    else:
        h = bytearray(HEADER_BYTES)
        struct.pack_into("<i", h, 0, HEADER_BYTES)                       # sizeof_hdr
        h[38] = ord("r")                                                  # regular (ANALYZE relic)
        struct.pack_into("<8f", h, 76, 1.0, *([1.0] * len(shape) + [1.0] * (7 - len(shape))))  # pixdim
        struct.pack_into("<f", h, 112, 1.0)                               # scl_slope
        struct.pack_into("<4f", h, 280, 1.0, 0.0, 0.0, 0.0)               # srow_x (unused: sform_code 0)
        struct.pack_into("<4f", h, 296, 0.0, 1.0, 0.0, 0.0)
        struct.pack_into("<4f", h, 312, 0.0, 0.0, 1.0, 0.0)
    struct.pack_into("<8h", h, 40, *dim)
    struct.pack_into("<hh", h, 70, code, bitpix)
    struct.pack_into("<f", h, 108, float(VOX_OFFSET))
    h[344:348] = b"n+1\0"                                                 # single-file magic
    return bytes(h)


def save(array, path, template=None):
    """Write `array` (indexed [x, y, z, t, ...]) to `path`; gzip when it ends in '.gz'."""
    arr = np.asarray(array)
    if arr.dtype == np.float16 or arr.dtype == np.bool_:
        arr = arr.astype(np.float32)
    blob = make_header(arr.shape, arr.dtype, template) + b"\0\0\0\0" + \
        np.asfortranarray(arr).astype(arr.dtype.newbyteorder("<"), copy=False).tobytes(order="F")
    if str(path).endswith(".gz"):
        with gzip.open(path, "wb", compresslevel=6) as f:
            f.write(blob)
    else:
        with open(path, "wb") as f:
            f.write(blob)


def load(path):
    """-> (array indexed [x, y, z, ...] with scl_slope/inter applied when set, header bytes)."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        blob = f.read()
    if len(blob) < VOX_OFFSET or struct.unpack_from("<i", blob, 0)[0] != HEADER_BYTES:
        raise ValueError(f"{path}: not a little-endian NIfTI-1 file")
    if blob[344:347] != b"n+1":
        raise ValueError(f"{path}: not a single-file NIfTI-1 image (magic {blob[344:348]!r})")
    dim = struct.unpack_from("<8h", blob, 40)
    code, _ = struct.unpack_from("<hh", blob, 70)
    if code not in _CODES:
        raise ValueError(f"{path}: unsupported datatype code {code}")
    shape = tuple(dim[1:1 + dim[0]])
    off = int(struct.unpack_from("<f", blob, 108)[0])
    dt = _CODES[code].newbyteorder("<")
    n = int(np.prod(shape))
    arr = np.frombuffer(blob, dt, n, off).reshape(shape, order="F")
    slope, inter = struct.unpack_from("<ff", blob, 112)
    if slope not in (0.0, 1.0) or inter != 0.0:
        if slope != 0.0:
            arr = arr * slope + inter
    return np.array(arr), blob[:HEADER_BYTES]
