"""ctypes binding to the HDF5 C library -- just what Keras' ``.h5`` checkpoints need (groups, contiguous numeric
datasets, string / string-array / numeric attributes).  ``h5py`` is not installable here; ``libhdf5`` itself is in the
image, and the C API below has been stable since 1.8.

Host-side file-format code: nothing here touches the GPU.  The library is looked up in this order: the
``LIPASR_HDF5_LIBRARY`` environment variable, the loader's search path, well-known install prefixes.  If none is
found every entry point raises -- there is no substitute format behind an ``.h5`` file name.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

_lib = None
hid_t = C.c_int64
hsize_t = C.c_ulonglong

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL, H5S_SCALAR = 0, 0, 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_CSET_ASCII, H5T_CSET_UTF8 = 0, 1
H5T_STR_NULLTERM, H5T_STR_NULLPAD, H5T_STR_SPACEPAD = 0, 1, 2
H5T_VARIABLE = C.c_size_t(-1).value

_SEARCH = ("libhdf5.so", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so",
           "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so", "/usr/local/lib/libhdf5.so", "/usr/lib64/libhdf5.so")


class HDF5Error(RuntimeError):
    pass


def _proto(lib, name, res, *args):
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, list(args)
    return fn


def library():
    """The loaded libhdf5 (ctypes.CDLL) with prototypes set; raises HDF5Error if the image has none."""
    global _lib, hid_t
    if _lib is not None:
        return _lib
    tried = []
    names = [os.environ.get("LIPASR_HDF5_LIBRARY"), ctypes.util.find_library("hdf5"), ctypes.util.find_library("hdf5_serial")]
    lib = None
    for cand in [n for n in names if n] + list(_SEARCH):
        try:
            lib = C.CDLL(cand)
            break
        except OSError as e:
            tried.append(f"{cand}: {e}")
    if lib is None:
        raise HDF5Error("libhdf5 not found (set LIPASR_HDF5_LIBRARY); an .h5 checkpoint cannot be read or written without it.\n  "
                        + "\n  ".join(tried))
    if lib.H5open() < 0:
        raise HDF5Error("H5open failed")
    maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
    lib.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
    lib.version = (maj.value, mnr.value, rel.value)
    hid_t = C.c_int64 if lib.version >= (1, 10, 0) else C.c_int  # hid_t grew to 64 bits in 1.10
    h, sz, I, p, s = hid_t, C.c_size_t, C.c_int, C.c_void_p, C.c_char_p
    _proto(lib, "H5Eset_auto2", I, h, p, p)(0, None, None)  # no error stack on stderr: failures raise here instead
    for name, res, args in (
        ("H5Fcreate", h, (s, C.c_uint, h, h)), ("H5Fopen", h, (s, C.c_uint, h)), ("H5Fclose", I, (h,)), ("H5Fflush", I, (h, I)),
        ("H5Gcreate2", h, (h, s, h, h, h)), ("H5Gclose", I, (h,)),
        ("H5Oopen", h, (h, s, h)), ("H5Oclose", I, (h,)), ("H5Lexists", I, (h, s, h)),
        ("H5Screate", h, (I,)), ("H5Screate_simple", h, (I, C.POINTER(hsize_t), C.POINTER(hsize_t))), ("H5Sclose", I, (h,)),
        ("H5Sget_simple_extent_ndims", I, (h,)), ("H5Sget_simple_extent_dims", I, (h, C.POINTER(hsize_t), C.POINTER(hsize_t))),
        ("H5Sget_simple_extent_npoints", C.c_longlong, (h,)),
        ("H5Dcreate2", h, (h, s, h, h, h, h, h)), ("H5Dopen2", h, (h, s, h)), ("H5Dclose", I, (h,)),
        ("H5Dget_space", h, (h,)), ("H5Dget_type", h, (h,)), ("H5Dwrite", I, (h, h, h, h, h, p)), ("H5Dread", I, (h, h, h, h, h, p)),
        ("H5Dvlen_reclaim", I, (h, h, h, p)),
        ("H5Acreate2", h, (h, s, h, h, h, h)), ("H5Aopen", h, (h, s, h)), ("H5Aexists", I, (h, s)), ("H5Aclose", I, (h,)),
        ("H5Aget_space", h, (h,)), ("H5Aget_type", h, (h,)), ("H5Awrite", I, (h, h, p)), ("H5Aread", I, (h, h, p)),
        ("H5Tcopy", h, (h,)), ("H5Tclose", I, (h,)), ("H5Tset_size", I, (h, sz)), ("H5Tget_size", sz, (h,)),
        ("H5Tset_strpad", I, (h, I)), ("H5Tset_cset", I, (h, I)), ("H5Tget_cset", I, (h,)), ("H5Tget_class", I, (h,)),
        ("H5Tget_sign", I, (h,)), ("H5Tis_variable_str", I, (h,)),
    ):
        _proto(lib, name, res, *args)
    lib.types = {k: hid_t.in_dll(lib, f"H5T_{k}_g").value for k in
                 ("NATIVE_FLOAT", "NATIVE_DOUBLE", "NATIVE_INT8", "NATIVE_UINT8", "NATIVE_INT16", "NATIVE_UINT16",
                  "NATIVE_INT32", "NATIVE_UINT32", "NATIVE_INT64", "NATIVE_UINT64", "C_S1")}
    _lib = lib
    return lib


_NP2H5 = {"float32": "NATIVE_FLOAT", "float64": "NATIVE_DOUBLE", "int8": "NATIVE_INT8", "uint8": "NATIVE_UINT8",
          "int16": "NATIVE_INT16", "uint16": "NATIVE_UINT16", "int32": "NATIVE_INT32", "uint32": "NATIVE_UINT32",
          "int64": "NATIVE_INT64", "uint64": "NATIVE_UINT64"}


def _ok(rc, what):
    if rc < 0:
        raise HDF5Error(f"HDF5: {what} failed")
    return rc


def _b(s):
    return s if isinstance(s, bytes) else str(s).encode("utf-8")


class File:
    """One open HDF5 file.  Paths are absolute-from-root strings ('model_weights/dense/dense/kernel:0')."""

    def __init__(self, path, mode="r"):
        self.L = library()
        self.path = str(path)
        if mode == "w":
            self.fid = self.L.H5Fcreate(_b(self.path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        elif mode == "r":
            if not os.path.exists(self.path):
                raise FileNotFoundError(self.path)
            self.fid = self.L.H5Fopen(_b(self.path), H5F_ACC_RDONLY, H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r' or 'w'")
        if self.fid < 0:
            raise HDF5Error(f"cannot open {self.path!r} as HDF5 (mode {mode!r})")

    def close(self):
        if self.fid >= 0:
            _ok(self.L.H5Fclose(self.fid), "H5Fclose")
            self.fid = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- objects
    def exists(self, path):
        """True if every link on the path exists (H5Lexists needs its parents to exist, so walk down)."""
        cur = ""
        for part in [p for p in path.split("/") if p]:
            cur = f"{cur}/{part}" if cur else part
            if self.L.H5Lexists(self.fid, _b(cur), H5P_DEFAULT) <= 0:
                return False
        return True

    def create_group(self, path):
        """mkdir -p."""
        cur = ""
        for part in [p for p in path.split("/") if p]:
            cur = f"{cur}/{part}" if cur else part
            if self.L.H5Lexists(self.fid, _b(cur), H5P_DEFAULT) > 0:
                continue
            g = _ok(self.L.H5Gcreate2(self.fid, _b(cur), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"H5Gcreate2({cur})")
            self.L.H5Gclose(g)

    def _open(self, path):
        return _ok(self.L.H5Oopen(self.fid, _b(path if path not in ("", "/") else "/"), H5P_DEFAULT), f"H5Oopen({path!r})")

    # ---- datasets
    def write_dataset(self, path, array):
        a = np.asarray(array, order="C")  # (ascontiguousarray would turn a 0-d value into shape (1,))
        if a.dtype.name not in _NP2H5:
            raise TypeError(f"dataset dtype {a.dtype} is not supported")
        parent = path.rsplit("/", 1)[0] if "/" in path else ""
        if parent:
            self.create_group(parent)
        t = self.L.types[_NP2H5[a.dtype.name]]
        if a.ndim == 0:
            sp = _ok(self.L.H5Screate(H5S_SCALAR), "H5Screate")
        else:
            dims = (hsize_t * a.ndim)(*a.shape)
            sp = _ok(self.L.H5Screate_simple(a.ndim, dims, None), "H5Screate_simple")
        try:
            d = _ok(self.L.H5Dcreate2(self.fid, _b(path), t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"H5Dcreate2({path})")
            try:
                if a.size:
                    _ok(self.L.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)), f"H5Dwrite({path})")
            finally:
                self.L.H5Dclose(d)
        finally:
            self.L.H5Sclose(sp)

    def _shape(self, sp):
        nd = _ok(self.L.H5Sget_simple_extent_ndims(sp), "H5Sget_simple_extent_ndims")
        if nd == 0:
            return ()
        dims = (hsize_t * nd)()
        _ok(self.L.H5Sget_simple_extent_dims(sp, dims, None), "H5Sget_simple_extent_dims")
        return tuple(int(x) for x in dims)

    def _numeric_dtype(self, t):
        cls, size = self.L.H5Tget_class(t), self.L.H5Tget_size(t)
        if cls == H5T_FLOAT and size in (4, 8):
            return np.dtype(f"float{8 * size}")
        if cls == H5T_INTEGER and size in (1, 2, 4, 8):
            return np.dtype(("int" if self.L.H5Tget_sign(t) else "uint") + str(8 * size))
        raise HDF5Error(f"unsupported HDF5 datatype (class {cls}, {size} bytes)")

    def read_dataset(self, path):
        d = _ok(self.L.H5Dopen2(self.fid, _b(path), H5P_DEFAULT), f"H5Dopen2({path})")
        try:
            sp, t = self.L.H5Dget_space(d), self.L.H5Dget_type(d)
            try:
                dt = self._numeric_dtype(t)
                out = np.empty(self._shape(sp), dtype=dt)
                if out.size:
                    _ok(self.L.H5Dread(d, self.L.types[_NP2H5[dt.name]], H5S_ALL, H5S_ALL, H5P_DEFAULT,
                                       out.ctypes.data_as(C.c_void_p)), f"H5Dread({path})")
                return out
            finally:
                self.L.H5Tclose(t)
                self.L.H5Sclose(sp)
        finally:
            self.L.H5Dclose(d)

    # ---- attributes
    def _str_type(self, size, utf8):
        t = _ok(self.L.H5Tcopy(self.L.types["C_S1"]), "H5Tcopy")
        _ok(self.L.H5Tset_size(t, size), "H5Tset_size")
        _ok(self.L.H5Tset_cset(t, H5T_CSET_UTF8 if utf8 else H5T_CSET_ASCII), "H5Tset_cset")
        if size != H5T_VARIABLE:
            _ok(self.L.H5Tset_strpad(t, H5T_STR_NULLPAD), "H5Tset_strpad")
        return t

    def write_attr(self, obj_path, name, value):
        """str / bytes -> scalar variable-length UTF-8 string (what h5py writes for a Python str);
        list of str / bytes -> 1-D array of fixed-length NUL-padded strings (h5py's numpy 'S' arrays), an EMPTY list ->
        a zero-length float64 array (np.asarray([]), which is what Keras ends up storing for weight-less layers);
        anything else -> numeric scalar / array."""
        o = self._open(obj_path)
        t_own = sp = None
        try:
            keep = None
            if isinstance(value, (str, bytes)):
                t = t_own = self._str_type(H5T_VARIABLE, True)
                sp = _ok(self.L.H5Screate(H5S_SCALAR), "H5Screate")
                keep = C.c_char_p(_b(value))
                buf = C.cast(C.pointer(keep), C.c_void_p)
            elif isinstance(value, (list, tuple)) and len(value) and isinstance(value[0], (str, bytes)):
                items = [_b(v) for v in value]
                width = max(1, max(len(v) for v in items))
                keep = np.array(items, dtype=f"S{width}")
                t = t_own = self._str_type(width, False)
                sp = _ok(self.L.H5Screate_simple(1, (hsize_t * 1)(len(items)), None), "H5Screate_simple")
                buf = keep.ctypes.data_as(C.c_void_p)
            else:
                keep = np.asarray(value, order="C")
                if keep.dtype.name not in _NP2H5:
                    raise TypeError(f"attribute {name!r}: dtype {keep.dtype} is not supported")
                t = self.L.types[_NP2H5[keep.dtype.name]]
                if keep.ndim == 0:
                    sp = _ok(self.L.H5Screate(H5S_SCALAR), "H5Screate")
                else:
                    sp = _ok(self.L.H5Screate_simple(keep.ndim, (hsize_t * keep.ndim)(*keep.shape), None), "H5Screate_simple")
                buf = keep.ctypes.data_as(C.c_void_p) if keep.size else None
            a = _ok(self.L.H5Acreate2(o, _b(name), t, sp, H5P_DEFAULT, H5P_DEFAULT), f"H5Acreate2({name})")
            try:
                if buf is not None:
                    _ok(self.L.H5Awrite(a, t, buf), f"H5Awrite({name})")
            finally:
                self.L.H5Aclose(a)
        finally:
            if sp is not None:
                self.L.H5Sclose(sp)
            if t_own is not None:
                self.L.H5Tclose(t_own)
            self.L.H5Oclose(o)

    def has_attr(self, obj_path, name):
        o = self._open(obj_path)
        try:
            return self.L.H5Aexists(o, _b(name)) > 0
        finally:
            self.L.H5Oclose(o)

    def read_attr(self, obj_path, name):
        """String attributes come back as str (scalar) or list of str (array); numeric ones as NumPy values."""
        o = self._open(obj_path)
        try:
            a = _ok(self.L.H5Aopen(o, _b(name), H5P_DEFAULT), f"H5Aopen({obj_path!r}, {name!r})")
            try:
                sp, t = self.L.H5Aget_space(a), self.L.H5Aget_type(a)
                try:
                    shape = self._shape(sp)
                    n = int(np.prod(shape)) if shape else 1
                    if self.L.H5Tget_class(t) == H5T_STRING:
                        if n == 0:
                            vals = []
                        elif self.L.H5Tis_variable_str(t) > 0:
                            mt = self._str_type(H5T_VARIABLE, self.L.H5Tget_cset(t) == H5T_CSET_UTF8)
                            try:
                                ptrs = (C.c_char_p * n)()
                                _ok(self.L.H5Aread(a, mt, C.cast(ptrs, C.c_void_p)), f"H5Aread({name})")
                                vals = [(p or b"").decode("utf-8") for p in ptrs]
                                self.L.H5Dvlen_reclaim(mt, sp, H5P_DEFAULT, C.cast(ptrs, C.c_void_p))
                            finally:
                                self.L.H5Tclose(mt)
                        else:
                            width = self.L.H5Tget_size(t)
                            raw = C.create_string_buffer(n * width)
                            _ok(self.L.H5Aread(a, t, C.cast(raw, C.c_void_p)), f"H5Aread({name})")
                            vals = [raw.raw[i * width:(i + 1) * width].split(b"\0", 1)[0].rstrip(b" ").decode("utf-8")
                                    for i in range(n)]
                        return vals[0] if shape == () else vals
                    dt = self._numeric_dtype(t)
                    out = np.empty(shape, dtype=dt)
                    if out.size:
                        _ok(self.L.H5Aread(a, self.L.types[_NP2H5[dt.name]], out.ctypes.data_as(C.c_void_p)), f"H5Aread({name})")
                    return out[()] if shape == () else out
                finally:
                    self.L.H5Tclose(t)
                    self.L.H5Sclose(sp)
            finally:
                self.L.H5Aclose(a)
        finally:
            self.L.H5Oclose(o)
