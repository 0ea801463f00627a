"""Minimal HDF5 reader/writer over libhdf5's C API (ctypes) for the reference's
dataset layout (datasets/dataset_parser.py:121-177):

    /<split>/inputs     1-D, variable-length float32 (T*F flattened, row-major (T,F)),
                        attribute ``num_feats`` (int64)
    /<split>/labels     1-D, variable-length UTF-8 strings
    /<split>/durations  1-D float32

The main interpreter of this image has no h5py; libhdf5.so (1.10.x) is loadable.
Host-side I/O only -- nothing here is on the measured path.  Files written by
h5py (the reference) are readable and vice versa (tests/test_h5lite.py checks both
directions against /opt/conda's h5py when it is present).
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int

_CANDIDATES = [os.environ.get('ASR_LIBHDF5'), '/opt/conda/lib/libhdf5.so',
               ctypes.util.find_library('hdf5'), 'libhdf5.so', 'libhdf5_serial.so',
               '/usr/lib/x86_64-linux-gnu/libhdf5_serial.so']
_lib = None


class H5Error(IOError):
    pass


class hvl_t(C.Structure):
    _fields_ = [('len', C.c_size_t), ('p', C.c_void_p)]


def available():
    try:
        _load()
        return True
    except H5Error:
        return False


def _load():
    global _lib
    if _lib is not None:
        return _lib
    last = None
    for cand in _CANDIDATES:
        if not cand:
            continue
        try:
            lib = C.CDLL(cand)
            break
        except OSError as e:
            last = e
    else:
        raise H5Error('libhdf5 not found (%s); set ASR_LIBHDF5 or use the .npz dataset '
                      'format' % last)
    sig = {
        'H5open': (herr_t, []),
        'H5Fcreate': (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        'H5Fopen': (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        'H5Fclose': (herr_t, [hid_t]),
        'H5Fis_hdf5': (C.c_int, [C.c_char_p]),
        'H5Gcreate2': (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
        'H5Gopen2': (hid_t, [hid_t, C.c_char_p, hid_t]),
        'H5Gclose': (herr_t, [hid_t]),
        'H5Lexists': (C.c_int, [hid_t, C.c_char_p, hid_t]),
        'H5Literate': (herr_t, [hid_t, C.c_int, C.c_int, C.POINTER(hsize_t), C.c_void_p,
                                C.c_void_p]),
        'H5Dcreate2': (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        'H5Dopen2': (hid_t, [hid_t, C.c_char_p, hid_t]),
        'H5Dclose': (herr_t, [hid_t]),
        'H5Dget_space': (hid_t, [hid_t]),
        'H5Dget_type': (hid_t, [hid_t]),
        'H5Dread': (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        'H5Dwrite': (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        'H5Dvlen_reclaim': (herr_t, [hid_t, hid_t, hid_t, C.c_void_p]),
        'H5Screate_simple': (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        'H5Screate': (hid_t, [C.c_int]),
        'H5Sclose': (herr_t, [hid_t]),
        'H5Sget_simple_extent_npoints': (C.c_int64, [hid_t]),
        'H5Sget_simple_extent_ndims': (C.c_int, [hid_t]),
        'H5Sget_simple_extent_dims': (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        'H5Aget_space': (hid_t, [hid_t]),
        'H5Tget_size': (C.c_size_t, [hid_t]),
        'H5Tis_variable_str': (C.c_int, [hid_t]),
        'H5Sselect_elements': (herr_t, [hid_t, C.c_int, C.c_size_t, C.POINTER(hsize_t)]),
        'H5Tvlen_create': (hid_t, [hid_t]),
        'H5Tcopy': (hid_t, [hid_t]),
        'H5Tset_size': (herr_t, [hid_t, C.c_size_t]),
        'H5Tset_cset': (herr_t, [hid_t, C.c_int]),
        'H5Tclose': (herr_t, [hid_t]),
        'H5Tget_class': (C.c_int, [hid_t]),
        'H5Acreate2': (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
        'H5Aopen': (hid_t, [hid_t, C.c_char_p, hid_t]),
        'H5Aexists': (C.c_int, [hid_t, C.c_char_p]),
        'H5Aread': (herr_t, [hid_t, hid_t, C.c_void_p]),
        'H5Awrite': (herr_t, [hid_t, hid_t, C.c_void_p]),
        'H5Aget_type': (hid_t, [hid_t]),
        'H5Aclose': (herr_t, [hid_t]),
        'H5Eset_auto2': (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.H5open()
    lib.H5Eset_auto2(0, None, None)            # no stderr spam; we raise instead
    _lib = lib
    return lib


def _g(name):
    return hid_t.in_dll(_load(), name).value


def _chk(v, what):
    if v < 0:
        raise H5Error('HDF5 call failed: %s' % what)
    return v


H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5S_SCALAR = 0
H5S_SELECT_SET = 0
H5T_CSET_UTF8 = 1
H5T_VARIABLE = C.c_size_t(-1).value
H5_INDEX_NAME, H5_ITER_NATIVE = 0, 2
H5T_FLOAT, H5T_STRING, H5T_VLEN = 1, 3, 9


def is_hdf5(fname):
    if not os.path.isfile(fname):
        return False
    try:
        return _load().H5Fis_hdf5(fname.encode()) > 0
    except H5Error:
        with open(fname, 'rb') as f:
            return f.read(8) == b'\x89HDF\r\n\x1a\n'


def _vlen_str_type():
    lib = _load()
    t = _chk(lib.H5Tcopy(_g('H5T_C_S1_g')), 'H5Tcopy')
    lib.H5Tset_size(t, H5T_VARIABLE)
    lib.H5Tset_cset(t, H5T_CSET_UTF8)
    return t


class Dataset(object):
    """1-D dataset; supports len(), integer / list / slice indexing (sorted lists like
    h5py), and .attrs[name] for integer/float/string attributes."""

    def __init__(self, parent_id, name):
        self.lib = _load()
        self.name = name
        self.id = _chk(self.lib.H5Dopen2(parent_id, name.encode(), H5P_DEFAULT), 'open ' + name)
        sp = self.lib.H5Dget_space(self.id)
        self.n = int(self.lib.H5Sget_simple_extent_npoints(sp))
        self.lib.H5Sclose(sp)
        t = self.lib.H5Dget_type(self.id)
        self.cls = self.lib.H5Tget_class(t)
        self.lib.H5Tclose(t)
        self.attrs = _Attrs(self.id)

    def __len__(self):
        return self.n

    @property
    def shape(self):
        return (self.n,)

    def _read(self, idx):
        lib = self.lib
        idx = np.asarray(idx, dtype=np.uint64).reshape(-1)
        k = len(idx)
        if k == 0:
            return []
        fsp = lib.H5Dget_space(self.id)
        coords = (hsize_t * k)(*[int(i) for i in idx])
        _chk(lib.H5Sselect_elements(fsp, H5S_SELECT_SET, k, coords), 'select')
        dims = (hsize_t * 1)(k)
        msp = lib.H5Screate_simple(1, dims, None)
        try:
            if self.cls == H5T_VLEN:
                mt = lib.H5Tvlen_create(_g('H5T_NATIVE_FLOAT_g'))
                buf = (hvl_t * k)()
                _chk(lib.H5Dread(self.id, mt, msp, fsp, H5P_DEFAULT, buf), 'read ' + self.name)
                out = []
                for e in buf:
                    if e.len:
                        arr = np.ctypeslib.as_array(C.cast(e.p, C.POINTER(C.c_float)),
                                                    shape=(e.len,)).copy()
                    else:
                        arr = np.zeros(0, np.float32)
                    out.append(arr)
                lib.H5Dvlen_reclaim(mt, msp, H5P_DEFAULT, buf)
                lib.H5Tclose(mt)
                return out
            if self.cls == H5T_STRING:
                mt = _vlen_str_type()
                buf = (C.c_char_p * k)()
                _chk(lib.H5Dread(self.id, mt, msp, fsp, H5P_DEFAULT, buf), 'read ' + self.name)
                out = [(b or b'').decode('utf-8') for b in buf]
                lib.H5Dvlen_reclaim(mt, msp, H5P_DEFAULT, buf)
                lib.H5Tclose(mt)
                return out
            out = np.empty(k, np.float32)
            _chk(lib.H5Dread(self.id, _g('H5T_NATIVE_FLOAT_g'), msp, fsp, H5P_DEFAULT,
                             out.ctypes.data_as(C.c_void_p)), 'read ' + self.name)
            return out
        finally:
            lib.H5Sclose(msp)
            lib.H5Sclose(fsp)

    def dims(self):
        """Full N-D shape (Keras weight datasets are N-D; the dataset files are 1-D)."""
        sp = self.lib.H5Dget_space(self.id)
        nd = self.lib.H5Sget_simple_extent_ndims(sp)
        d = (hsize_t * max(nd, 1))()
        if nd > 0:
            self.lib.H5Sget_simple_extent_dims(sp, d, None)
        self.lib.H5Sclose(sp)
        return tuple(int(d[i]) for i in range(nd))

    def read_array(self):
        """The whole numeric dataset as a float32 ndarray of its N-D shape."""
        out = np.empty(self.n, np.float32)
        if self.n:
            _chk(self.lib.H5Dread(self.id, _g('H5T_NATIVE_FLOAT_g'), H5S_ALL, H5S_ALL, H5P_DEFAULT,
                                  out.ctypes.data_as(C.c_void_p)), 'read ' + self.name)
        return out.reshape(self.dims())

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            k = int(key)
            if k < 0:
                k += self.n
            if not 0 <= k < self.n:
                raise IndexError(key)
            return self._read([k])[0]
        if isinstance(key, slice):
            return self._read(list(range(*key.indices(self.n))))
        key = list(key)
        if any(b <= a for a, b in zip(key, key[1:])):
            raise TypeError('Indexing elements must be in increasing order')   # as h5py
        return self._read(key)

    def close(self):
        if self.id:
            self.lib.H5Dclose(self.id)
            self.id = 0


class _Attrs(object):
    def __init__(self, obj_id):
        self.obj = obj_id
        self.lib = _load()

    def __contains__(self, name):
        return self.lib.H5Aexists(self.obj, name.encode()) > 0

    def keys(self):
        return [k for k in ('num_feats', 'training_args') if k in self]

    def __getitem__(self, name):
        lib = self.lib
        if name not in self:
            raise KeyError(name)
        a = _chk(lib.H5Aopen(self.obj, name.encode(), H5P_DEFAULT), 'attr ' + name)
        t = lib.H5Aget_type(a)
        cls = lib.H5Tget_class(t)
        try:
            if cls == H5T_STRING:          # read with the file's own string type
                if lib.H5Tis_variable_str(t) > 0:
                    buf = C.c_char_p()
                    _chk(lib.H5Aread(a, t, C.byref(buf)), 'attr read')
                    return (buf.value or b'').decode('utf-8')
                raw = C.create_string_buffer(int(lib.H5Tget_size(t)) + 1)
                _chk(lib.H5Aread(a, t, raw), 'attr read')
                return raw.raw.split(b'\0')[0].decode('utf-8')
            if cls == H5T_FLOAT:
                v = C.c_double()
                _chk(lib.H5Aread(a, _g('H5T_NATIVE_DOUBLE_g'), C.byref(v)), 'attr read')
                return v.value
            v = C.c_int64()
            _chk(lib.H5Aread(a, _g('H5T_NATIVE_INT64_g'), C.byref(v)), 'attr read')
            return v.value
        finally:
            lib.H5Tclose(t)
            lib.H5Aclose(a)

    def get_strings(self, name):
        """An attribute holding an ARRAY of strings (Keras' layer_names / weight_names,
        stored by h5py as fixed-length byte strings) -> list of str."""
        lib = self.lib
        if name not in self:
            raise KeyError(name)
        a = _chk(lib.H5Aopen(self.obj, name.encode(), H5P_DEFAULT), 'attr ' + name)
        sp = lib.H5Aget_space(a)
        n = int(lib.H5Sget_simple_extent_npoints(sp))
        lib.H5Sclose(sp)
        ft = lib.H5Aget_type(a)
        try:
            if lib.H5Tis_variable_str(ft) > 0:
                buf = (C.c_char_p * max(n, 1))()      # read with the file's own type:
                _chk(lib.H5Aread(a, ft, buf), 'attr read ' + name)   # no cset conversion
                return [(buf[i] or b'').decode('utf-8') for i in range(n)]
            width = int(lib.H5Tget_size(ft))            # fixed-length, NUL padded
            raw = C.create_string_buffer(max(n * width, 1))
            _chk(lib.H5Aread(a, ft, raw), 'attr read ' + name)
            return [raw.raw[i * width:(i + 1) * width].split(b'\0')[0].decode('utf-8')
                    for i in range(n)]
        finally:
            lib.H5Tclose(ft)
            lib.H5Aclose(a)

    def set_strings(self, name, strings):
        """Writes a 1-D array-of-strings attribute (fixed-length bytes, as h5py does for a
        numpy 'S' array)."""
        lib = self.lib
        enc = [s.encode('utf-8') for s in strings]
        width = max([len(e) for e in enc] + [1])
        t = _chk(lib.H5Tcopy(_g('H5T_C_S1_g')), 'H5Tcopy')
        lib.H5Tset_size(t, width)
        dims = (hsize_t * 1)(len(enc))
        sp = lib.H5Screate_simple(1, dims, None)
        a = _chk(lib.H5Acreate2(self.obj, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT),
                 'attr create')
        raw = b''.join(e.ljust(width, b'\0') for e in enc)
        buf = C.create_string_buffer(raw, len(raw) or 1)
        _chk(lib.H5Awrite(a, t, buf), 'attr write')
        lib.H5Aclose(a)
        lib.H5Sclose(sp)
        lib.H5Tclose(t)

    def __setitem__(self, name, value):
        lib = self.lib
        sp = lib.H5Screate(H5S_SCALAR)
        if isinstance(value, str):
            t = _vlen_str_type()
            a = _chk(lib.H5Acreate2(self.obj, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT),
                     'attr create')
            buf = C.c_char_p(value.encode('utf-8'))
            _chk(lib.H5Awrite(a, t, C.byref(buf)), 'attr write')
            lib.H5Tclose(t)
        elif isinstance(value, float):
            a = _chk(lib.H5Acreate2(self.obj, name.encode(), _g('H5T_IEEE_F64LE_g'), sp,
                                    H5P_DEFAULT, H5P_DEFAULT), 'attr create')
            v = C.c_double(value)
            _chk(lib.H5Awrite(a, _g('H5T_NATIVE_DOUBLE_g'), C.byref(v)), 'attr write')
        else:
            a = _chk(lib.H5Acreate2(self.obj, name.encode(), _g('H5T_STD_I64LE_g'), sp,
                                    H5P_DEFAULT, H5P_DEFAULT), 'attr create')
            v = C.c_int64(int(value))
            _chk(lib.H5Awrite(a, _g('H5T_NATIVE_INT64_g'), C.byref(v)), 'attr write')
        lib.H5Aclose(a)
        lib.H5Sclose(sp)


class Group(object):
    def __init__(self, gid, owns=True):
        self.lib = _load()
        self.id = gid
        self.owns = owns
        self.attrs = _Attrs(gid)
        self._children = []

    def __contains__(self, name):
        return self.lib.H5Lexists(self.id, name.strip('/').encode(), H5P_DEFAULT) > 0

    def __getitem__(self, name):
        name = name.strip('/')
        if name == '':
            return self
        if name not in self:
            raise KeyError(name)
        gid = self.lib.H5Gopen2(self.id, name.encode(), H5P_DEFAULT)
        if gid >= 0:
            child = Group(gid)
        else:
            child = Dataset(self.id, name)
        self._children.append(child)
        return child

    def keys(self):
        names = []
        CB = C.CFUNCTYPE(herr_t, hid_t, C.c_char_p, C.c_void_p, C.c_void_p)

        def cb(g, name, info, data):
            names.append(name.decode())
            return 0
        keep = CB(cb)
        idx = hsize_t(0)
        self.lib.H5Literate(self.id, H5_INDEX_NAME, H5_ITER_NATIVE, C.byref(idx),
                            C.cast(keep, C.c_void_p), None)
        return names

    def create_group(self, name):
        gid = _chk(self.lib.H5Gcreate2(self.id, name.encode(), H5P_DEFAULT, H5P_DEFAULT,
                                       H5P_DEFAULT), 'create group ' + name)
        child = Group(gid)
        self._children.append(child)
        return child

    def _create(self, name, file_type, n):
        lib = self.lib
        dims = (hsize_t * 1)(n)
        sp = lib.H5Screate_simple(1, dims, None)
        d = _chk(lib.H5Dcreate2(self.id, name.encode(), file_type, sp, H5P_DEFAULT, H5P_DEFAULT,
                                H5P_DEFAULT), 'create dataset ' + name)
        lib.H5Sclose(sp)
        return d

    def write_vlen_float(self, name, arrays, attrs=None):
        """1-D dataset of variable-length float32 vectors."""
        lib = self.lib
        n = len(arrays)
        keep = [np.ascontiguousarray(a, dtype=np.float32).reshape(-1) for a in arrays]
        buf = (hvl_t * n)()
        for i, a in enumerate(keep):
            buf[i].len = a.size
            buf[i].p = a.ctypes.data
        ft = lib.H5Tvlen_create(_g('H5T_IEEE_F32LE_g'))
        mt = lib.H5Tvlen_create(_g('H5T_NATIVE_FLOAT_g'))
        d = self._create(name, ft, n)
        _chk(lib.H5Dwrite(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf), 'write ' + name)
        for k, v in (attrs or {}).items():
            _Attrs(d)[k] = v
        lib.H5Dclose(d)
        lib.H5Tclose(ft)
        lib.H5Tclose(mt)

    def write_strings(self, name, strings):
        lib = self.lib
        n = len(strings)
        enc = [s.encode('utf-8') for s in strings]
        buf = (C.c_char_p * n)(*enc)
        t = _vlen_str_type()
        d = self._create(name, t, n)
        _chk(lib.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf), 'write ' + name)
        lib.H5Dclose(d)
        lib.H5Tclose(t)

    def write_array(self, name, array):
        """N-D float32 dataset (Keras weight layout)."""
        lib = self.lib
        a = np.ascontiguousarray(array, dtype=np.float32)
        shape = a.shape if a.ndim else (1,)
        dims = (hsize_t * len(shape))(*shape)
        sp = lib.H5Screate_simple(len(shape), dims, None)
        d = _chk(lib.H5Dcreate2(self.id, name.encode(), _g('H5T_IEEE_F32LE_g'), sp, H5P_DEFAULT,
                                H5P_DEFAULT, H5P_DEFAULT), 'create dataset ' + name)
        lib.H5Sclose(sp)
        _chk(lib.H5Dwrite(d, _g('H5T_NATIVE_FLOAT_g'), H5S_ALL, H5S_ALL, H5P_DEFAULT,
                          a.ctypes.data_as(C.c_void_p)), 'write ' + name)
        lib.H5Dclose(d)

    def write_float(self, name, values):
        lib = self.lib
        a = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
        d = self._create(name, _g('H5T_IEEE_F32LE_g'), a.size)
        _chk(lib.H5Dwrite(d, _g('H5T_NATIVE_FLOAT_g'), H5S_ALL, H5S_ALL, H5P_DEFAULT,
                          a.ctypes.data_as(C.c_void_p)), 'write ' + name)
        lib.H5Dclose(d)

    def close(self):
        for c in self._children:
            c.close()
        self._children = []
        if self.id and self.owns:
            self.lib.H5Gclose(self.id)
        self.id = 0


class File(Group):
    def __init__(self, fname, mode='r'):
        lib = _load()
        if mode == 'r':
            fid = lib.H5Fopen(fname.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode in ('r+', 'a') and os.path.exists(fname):
            fid = lib.H5Fopen(fname.encode(), H5F_ACC_RDWR, H5P_DEFAULT)
        else:
            fid = lib.H5Fcreate(fname.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        _chk(fid, 'open %s' % fname)
        self.fid = fid
        root = _chk(lib.H5Gopen2(fid, b'/', H5P_DEFAULT), 'open root')
        super(File, self).__init__(root)

    def close(self):
        super(File, self).close()
        if self.fid:
            self.lib.H5Fclose(self.fid)
            self.fid = 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
