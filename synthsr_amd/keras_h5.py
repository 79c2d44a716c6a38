"""Keras `.h5` weight files read without an HDF5 library (none is installed next to this package's interpreter).

The reference keeps its networks as Keras HDF5 files: `KC.ModelCheckpoint(save_file_name)` writes one per epoch
(SynthSR/training.py:430), `model.load_weights(checkpoint, by_name=True)` resumes from one (SynthSR/training.py:363), and the
released models `models/SynthSR_v10_210712*.h5` are loaded the same way (scripts/predict_command_line.py:79-82,
scripts/predict_command_line_hyperfine.py:73).  `load_keras_weights()` returns such a file as the `{name: ndarray}`
state dict of `synthsr_amd.unet.UNet3D` / `training.save_checkpoint` (Keras weight names minus the `:0` suffix).

`H5File` is a small reader of the HDF5 file format (HDF5 File Format Specification, version 3.0): superblock v0-v3,
version-1 and version-2 object headers, symbol-table groups (v1 B-tree + local heap) and compact link messages,
contiguous / compact / chunked (v1 B-tree) layouts with the deflate, shuffle and fletcher32 filters, fixed-point /
floating-point / fixed-length-string / variable-length-string datatypes, attributes v1-v3.  Not read: dense link or
attribute storage (fractal heaps; only produced with libver='latest' and many entries), the version-4 chunk indices,
compound / enum / reference types - those raise `H5FormatError`, nothing is guessed.

Pinned by tests/test_keras_h5.py against files written with the real library (tests/golden/gen/make_keras_h5.py)."""
import struct
import zlib

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'


class H5FormatError(ValueError):
    pass


def _pad8(n):
    return (n + 7) & ~7


class _Type:
    """decoded datatype message: kind in ('num', 'str', 'vstr', 'other')"""

    def __init__(self, kind, size, dtype=None, length=0):
        self.kind, self.size, self.dtype, self.length = kind, size, dtype, length


class H5File:
    def __init__(self, path):
        self.path = path
        with open(path, 'rb') as f:
            self.buf = f.read()
        base = 0
        while self.buf[base:base + 8] != SIGNATURE:        # the superblock may sit at 0, 512, 1024, 2048, ...
            base = 512 if base == 0 else base * 2
            if base + 8 > len(self.buf):
                raise H5FormatError('%s is not an HDF5 file' % path)
        b = self.buf
        ver = b[base + 8]
        if ver in (0, 1):
            self.so, self.sl = b[base + 13], b[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            self.base = self._uint(p, self.so)
            p += 4 * self.so                                 # base, free-space, end-of-file, driver-info addresses
            root = self._uint(p + self.so, self.so)          # root symbol-table entry: name offset, header address
        elif ver in (2, 3):
            self.so, self.sl = b[base + 9], b[base + 10]
            p = base + 12
            self.base = self._uint(p, self.so)
            root = self._uint(p + 3 * self.so, self.so)
        else:
            raise H5FormatError('unsupported HDF5 superblock version %d' % ver)
        if self.so not in (4, 8) or self.sl not in (4, 8):
            raise H5FormatError('unsupported offset/length sizes %d/%d' % (self.so, self.sl))
        self.undef = (1 << (8 * self.so)) - 1
        self.root = Group(self, root, '/')

    # ------------------------------------------------------------------ primitives
    def _uint(self, off, n):
        return int.from_bytes(self.buf[off:off + n], 'little')

    def _addr(self, off):
        a = self._uint(off, self.so)
        return a if a == self.undef else a + self.base

    def _messages(self, addr):
        """[(type, flags, data offset, data size)] of the object header at addr, continuation blocks followed"""
        b = self.buf
        msgs = []
        if b[addr:addr + 4] == b'OHDR':
            if b[addr + 4] != 2:
                raise H5FormatError('object header version %d' % b[addr + 4])
            flags = b[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            n = 1 << (flags & 3)
            size = self._uint(p, n)
            p += n
            blocks = [(p, size)]
            order = 2 if flags & 0x04 else 0
            while blocks:
                p, size = blocks.pop(0)
                end = p + size
                while p + 4 + order <= end:
                    t, sz, fl = b[p], self._uint(p + 1, 2), b[p + 3]
                    d = p + 4 + order
                    if t == 0x10:
                        blocks.append((self._addr(d) + 4, self._uint(d + self.so, self.sl) - 8))   # OCHK sig, checksum
                    elif t != 0:
                        msgs.append((t, fl, d, sz))
                    p = d + sz
            return msgs
        if b[addr] != 1:
            raise H5FormatError('no object header at %d' % addr)
        nmsg = self._uint(addr + 2, 2)
        blocks = [(addr + 16, self._uint(addr + 8, 4))]
        seen = 0
        while blocks and seen < nmsg:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and seen < nmsg:
                t, sz, fl = self._uint(p, 2), self._uint(p + 2, 2), b[p + 4]
                d = p + 8
                seen += 1
                if t == 0x10:
                    blocks.append((self._addr(d), self._uint(d + self.so, self.sl)))
                elif t != 0:
                    msgs.append((t, fl, d, sz))
                p = d + sz
        return msgs

    # ------------------------------------------------------------------ message decoders
    def _dataspace(self, off):
        b = self.buf
        ver, rank, flags = b[off], b[off + 1], b[off + 2]
        if ver == 1:
            p = off + 8
        elif ver == 2:
            if b[off + 3] == 2:
                return None                                   # null dataspace: no elements
            p = off + 4
        else:
            raise H5FormatError('dataspace message version %d' % ver)
        return tuple(self._uint(p + i * self.sl, self.sl) for i in range(rank))

    def _datatype(self, off):
        b = self.buf
        cls, bits, size = b[off] & 15, b[off + 1] | (b[off + 2] << 8) | (b[off + 3] << 16), self._uint(off + 4, 4)
        order = '>' if bits & 1 else '<'
        if cls == 0:
            return _Type('num', size, np.dtype('%s%s%d' % (order, 'i' if bits & 8 else 'u', size)))
        if cls == 1:
            if size not in (2, 4, 8):
                raise H5FormatError('floating-point type of %d bytes' % size)
            return _Type('num', size, np.dtype('%sf%d' % (order, size)))
        if cls == 3:
            return _Type('str', size, np.dtype('S%d' % size))
        if cls == 9 and (bits & 15) == 1:
            return _Type('vstr', 4 + self.so + 4)
        return _Type('other', size)

    def _global_heap_object(self, addr, index):
        b = self.buf
        if b[addr:addr + 4] != b'GCOL':
            raise H5FormatError('no global heap collection at %d' % addr)
        end = addr + self._uint(addr + 8, self.sl)
        p = addr + 8 + self.sl
        while p + 8 + self.sl <= end:
            idx, size = self._uint(p, 2), self._uint(p + 8, self.sl)
            if idx == index:
                return b[p + 8 + self.sl:p + 8 + self.sl + size]
            if idx == 0:
                break
            p += 8 + self.sl + _pad8(size)
        raise H5FormatError('global heap object %d not found' % index)

    def _decode(self, raw, typ, shape):
        n = 0 if shape is None else int(np.prod(shape, dtype=np.int64))
        shape = (0,) if shape is None else shape
        if typ.kind in ('num', 'str'):
            return np.frombuffer(raw, dtype=typ.dtype, count=n).reshape(shape).copy()
        if typ.kind == 'vstr':
            out = []
            for i in range(n):
                p = i * typ.size
                length = int.from_bytes(raw[p:p + 4], 'little')
                addr = int.from_bytes(raw[p + 4:p + 4 + self.so], 'little')
                index = int.from_bytes(raw[p + 4 + self.so:p + 8 + self.so], 'little')
                out.append(b'' if length == 0 or addr in (0, self.undef) else
                           self._global_heap_object(addr + self.base, index)[:length])
            return np.array(out, dtype=object).reshape(shape)
        raise H5FormatError('unsupported datatype class')

    def _attribute(self, off):
        b = self.buf
        ver = b[off]
        nsz, tsz, ssz = self._uint(off + 2, 2), self._uint(off + 4, 2), self._uint(off + 6, 2)
        if ver == 1:
            p = off + 8
            name = b[p:p + nsz]
            p += _pad8(nsz)
            t_off = p
            p += _pad8(tsz)
            s_off = p
            p += _pad8(ssz)
        elif ver in (2, 3):
            if b[off + 1] & 3:
                raise H5FormatError('attribute with a shared datatype/dataspace')
            p = off + (8 if ver == 2 else 9)
            name = b[p:p + nsz]
            t_off = p + nsz
            s_off = t_off + tsz
            p = s_off + ssz
        else:
            raise H5FormatError('attribute message version %d' % ver)
        name = name.split(b'\0', 1)[0].decode('utf8')
        typ, shape = self._datatype(t_off), self._dataspace(s_off)
        n = 0 if shape is None else int(np.prod(shape, dtype=np.int64))
        if typ.kind == 'other':
            return name, None
        val = self._decode(b[p:p + n * typ.size], typ, shape)
        return name, (val[()] if shape == () else val)

    def attributes(self, addr):
        out = {}
        for t, fl, d, sz in self._messages(addr):
            if t == 0x0C:
                if fl & 2:
                    raise H5FormatError('shared attribute message')
                k, v = self._attribute(d)
                out[k] = v
            elif t == 0x15 and sz >= 2:                       # attribute info: dense storage when the heap address is set
                p = d + 2 + (2 if self.buf[d + 1] & 1 else 0)
                if self._uint(p, self.so) != self.undef:
                    raise H5FormatError('densely stored attributes (fractal heap) are not supported')
        return out

    # ------------------------------------------------------------------ groups
    def links(self, addr):
        """{name: object header address} of the group at addr"""
        b = self.buf
        out = {}
        for t, fl, d, sz in self._messages(addr):
            if t == 0x11:
                heap = self._addr(d + self.so)
                if b[heap:heap + 4] != b'HEAP':
                    raise H5FormatError('no local heap at %d' % heap)
                self._group_btree(self._addr(d), self._addr(heap + 8 + 2 * self.sl), out)
            elif t == 0x06:
                flags = b[d + 1]
                p = d + 2
                ltype = 0
                if flags & 0x08:
                    ltype = b[p]
                    p += 1
                if flags & 0x04:
                    p += 8
                if flags & 0x10:
                    p += 1
                n = 1 << (flags & 3)
                ln = self._uint(p, n)
                p += n
                name = b[p:p + ln].decode('utf8')
                if ltype == 0:
                    out[name] = self._addr(p + ln)
            elif t == 0x02:
                p = d + 2 + (8 if b[d + 1] & 1 else 0)
                if self._uint(p, self.so) != self.undef:
                    raise H5FormatError('densely stored links (fractal heap, libver="latest" with many members) are '
                                        'not supported')
        return out

    def _group_btree(self, addr, heap_data, out):
        b = self.buf
        if b[addr:addr + 4] != b'TREE' or b[addr + 4] != 0:
            raise H5FormatError('no group B-tree node at %d' % addr)
        level, n = b[addr + 5], self._uint(addr + 6, 2)
        p = addr + 8 + 2 * self.so + self.sl
        for _ in range(n):
            child = self._addr(p)
            p += self.so + self.sl
            if level > 0:
                self._group_btree(child, heap_data, out)
                continue
            if b[child:child + 4] != b'SNOD':
                raise H5FormatError('no symbol-table node at %d' % child)
            q = child + 8
            for _ in range(self._uint(child + 6, 2)):
                name_off, obj, cache = self._uint(q, self.so), self._addr(q + self.so), self._uint(q + 2 * self.so, 4)
                q += 2 * self.so + 24
                s = heap_data + name_off
                name = b[s:b.index(b'\0', s)].decode('utf8')
                if cache != 2:                                 # 2: symbolic link (no object header)
                    out[name] = obj
        return out

    # ------------------------------------------------------------------ datasets
    def dataset_info(self, addr):
        shape = typ = layout = None
        filters = []
        is_dataset = False
        b = self.buf
        for t, fl, d, sz in self._messages(addr):
            if t in (1, 3, 8, 0x0B) and fl & 2:
                raise H5FormatError('shared (committed) datatype/dataspace messages are not supported')
            if t == 1:
                shape = self._dataspace(d)
            elif t == 3:
                typ = self._datatype(d)
            elif t == 8:
                is_dataset = True
                ver, cls = b[d], b[d + 1]
                if ver not in (3, 4):
                    raise H5FormatError('data layout message version %d' % ver)
                if cls == 0:
                    layout = ('compact', d + 4, self._uint(d + 2, 2))
                elif cls == 1:
                    layout = ('contiguous', self._addr(d + 2), self._uint(d + 2 + self.so, self.sl))
                elif cls == 2 and ver == 3:
                    nd = b[d + 2]
                    dims = tuple(self._uint(d + 3 + self.so + 4 * i, 4) for i in range(nd))
                    layout = ('chunked', self._addr(d + 3), dims)
                else:
                    raise H5FormatError('data layout class %d of message version %d is not supported' % (cls, ver))
            elif t == 0x0B:
                ver, nf = b[d], b[d + 1]
                p = d + (8 if ver == 1 else 2)
                for _ in range(nf):
                    fid = self._uint(p, 2)
                    p += 2
                    nlen = 0
                    if ver == 1 or fid >= 256:
                        nlen = self._uint(p, 2)
                        p += 2
                    ncd = self._uint(p + 2, 2)
                    p += 4 + (_pad8(nlen) if ver == 1 else nlen)
                    cd = [self._uint(p + 4 * i, 4) for i in range(ncd)]
                    p += 4 * ncd + (4 if ver == 1 and ncd % 2 else 0)
                    filters.append((fid, cd))
        if not is_dataset:
            return None
        return shape, typ, layout, filters

    def read_dataset(self, addr):
        info = self.dataset_info(addr)
        if info is None:
            raise H5FormatError('object at %d is not a dataset' % addr)
        shape, typ, layout, filters = info
        if typ.kind == 'other':
            raise H5FormatError('unsupported datatype class')
        if shape is None:
            return self._decode(b'', typ, None)
        n = int(np.prod(shape, dtype=np.int64))
        if layout[0] == 'compact':
            return self._decode(self.buf[layout[1]:layout[1] + n * typ.size], typ, shape)
        if layout[0] == 'contiguous':
            if layout[1] == self.undef:                        # never written: fill value (zeros)
                return self._decode(bytes(n * typ.size), typ, shape)
            return self._decode(self.buf[layout[1]:layout[1] + n * typ.size], typ, shape)
        # chunked: v1 B-tree of (size, filter mask, offsets) -> chunk address
        btree, cdims = layout[1], layout[2][:-1]
        if typ.kind == 'vstr':
            raise H5FormatError('chunked variable-length datasets are not supported')
        out = np.zeros(shape, dtype=typ.dtype)
        if btree != self.undef:
            for offs, caddr, csize, mask in self._chunks(btree, len(cdims)):
                raw = self.buf[caddr:caddr + csize]
                for i in range(len(filters) - 1, -1, -1):
                    if mask & (1 << i):
                        continue
                    fid, cd = filters[i]
                    if fid == 1:
                        raw = zlib.decompress(raw)
                    elif fid == 2:
                        es = cd[0] if cd else typ.size
                        a = np.frombuffer(raw, np.uint8)
                        m = a.size // es
                        raw = np.concatenate([a[:m * es].reshape(es, m).T.reshape(-1), a[m * es:]]).tobytes()
                    elif fid == 3:
                        raw = raw[:-4]
                    else:
                        raise H5FormatError('unsupported HDF5 filter %d' % fid)
                chunk = np.frombuffer(raw, dtype=typ.dtype, count=int(np.prod(cdims))).reshape(cdims)
                src = tuple(slice(0, min(c, s - o)) for c, s, o in zip(cdims, shape, offs))
                dst = tuple(slice(o, min(o + c, s)) for c, s, o in zip(cdims, shape, offs))
                out[dst] = chunk[src]
        return out

    def _chunks(self, addr, nd):
        b = self.buf
        if b[addr:addr + 4] != b'TREE' or b[addr + 4] != 1:
            raise H5FormatError('no chunk B-tree node at %d' % addr)
        level, n = b[addr + 5], self._uint(addr + 6, 2)
        ksz = 8 + 8 * (nd + 1)
        p = addr + 8 + 2 * self.so
        for _ in range(n):
            csize, mask = self._uint(p, 4), self._uint(p + 4, 4)
            offs = tuple(self._uint(p + 8 + 8 * i, 8) for i in range(nd))
            child = self._addr(p + ksz)
            p += ksz + self.so
            if level > 0:
                yield from self._chunks(child, nd)
            else:
                yield offs, child, csize, mask

    def __getitem__(self, path):
        return self.root[path]


class Group:
    def __init__(self, file, addr, name):
        self.file, self.addr, self.name = file, addr, name
        self._links = None
        self._attrs = None

    @property
    def attrs(self):
        if self._attrs is None:
            self._attrs = self.file.attributes(self.addr)
        return self._attrs

    def _members(self):
        if self._links is None:
            self._links = self.file.links(self.addr)
        return self._links

    def keys(self):
        return sorted(self._members())

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split('/') if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            members = node._members()
            if part not in members:
                raise KeyError('%s (in %s of %s)' % (path, node.name, self.file.path))
            addr = members[part]
            child_name = node.name.rstrip('/') + '/' + part
            node = (Group(self.file, addr, child_name) if self.file.dataset_info(addr) is None
                    else Dataset(self.file, addr, child_name))
        return node


class Dataset:
    def __init__(self, file, addr, name):
        self.file, self.addr, self.name = file, addr, name
        info = file.dataset_info(addr)
        self.shape = info[0]
        self.dtype = info[1].dtype if info[1].kind in ('num', 'str') else object

    @property
    def attrs(self):
        return self.file.attributes(self.addr)

    def read(self):
        return self.file.read_dataset(self.addr)


# ---------------------------------------------------------------------- Keras layer on top
def _text(x):
    return x.decode('utf8') if isinstance(x, (bytes, np.bytes_)) else str(x)


def _string_list(attrs, name):
    """Keras' load_attributes_from_hdf5_group: `name`, or the chunks `name0`, `name1`, ... of a long list"""
    if name in attrs:
        return [_text(n) for n in np.asarray(attrs[name]).reshape(-1)]
    out, i = [], 0
    while '%s%d' % (name, i) in attrs:
        out.extend(_text(n) for n in np.asarray(attrs['%s%d' % (name, i)]).reshape(-1))
        i += 1
    if i == 0:
        raise H5FormatError('attribute %r not found: not a Keras weights file?' % name)
    return out


def load_keras_weights(path):
    """{'<layer>/<weight>': float32 ndarray} of a Keras `save_weights()` file or of a full `model.save()` /
    `ModelCheckpoint` file (weights under /model_weights; the optimizer slots: load_keras_optimizer)."""
    f = H5File(path)
    g = f.root
    if 'layer_names' not in g.attrs and 'layer_names0' not in g.attrs and 'model_weights' in g:
        g = g['model_weights']
    out = {}
    for layer in _string_list(g.attrs, 'layer_names'):
        lg = g[layer]
        names = _string_list(lg.attrs, 'weight_names') if ('weight_names' in lg.attrs or
                                                           'weight_names0' in lg.attrs) else []
        for wn in names:
            arr = lg[wn].read()
            key = wn.rsplit(':', 1)[0] if ':' in wn.rsplit('/', 1)[-1] else wn
            if '/' not in key:
                key = layer + '/' + key
            out[key] = np.ascontiguousarray(arr, dtype=np.float32)
    return out


def _adam_slot_count(names, arrays):
    """number of trainable weights behind Adam's slot list, decided FROM THE DATA (a count divisible by both 2 and 3 is
    ambiguous): [ms, vs, vhats] when the last third is what Keras 2.3.1 writes for amsgrad=False -- one shape-(1,) zero
    placeholder per weight (keras/optimizers.py: `vhats = [K.zeros(1) for _ in params]`) -- or, amsgrad=True, arrays shaped
    like the first third, with 'vhat' in their names if the names say anything; otherwise [ms, vs] with the two halves shaped
    alike.  None if neither reading fits."""
    k = len(arrays)
    shapes = [tuple(np.shape(a)) for a in arrays]
    if k % 3 == 0 and k:
        n = k // 3
        tail = shapes[2 * n:]
        placeholders = all(t == (1,) for t in tail) and any(sh != (1,) for sh in shapes[:n])
        named = all('vhat' in str(nm).lower() for nm in names[2 * n:])
        if shapes[:n] == shapes[n:2 * n] and (placeholders or (named and tail == shapes[:n])):
            return n
    if k % 2 == 0 and k:
        n = k // 2
        if shapes[:n] == shapes[n:]:
            return n
    if k % 3 == 0 and k and shapes[:k // 3] == shapes[k // 3:2 * k // 3] == shapes[2 * k // 3:]:
        return k // 3       # three alike thirds that are not two alike halves: amsgrad slots under foreign names
    return None


def load_keras_optimizer(path):
    """Adam slots of a full-model Keras file (`model.save()` / `ModelCheckpoint`, what SynthSR/training.py:429-439 resumes from
    with `models.load_model`): -> (iterations, [m_i], [v_i]) or None if the file has no /optimizer_weights.  Keras 2.3.1's
    `Adam.weights` is [iterations] + ms + vs + vhats with one entry per trainable weight of the model, IN THE ORDER OF
    `model.trainable_weights` (keras/optimizers.py, third party: restated); the variable names differ between Keras / TF
    versions, so the slots are taken by position and the caller checks them against its own shapes."""
    f = H5File(path)
    if 'optimizer_weights' not in f.root:
        return None
    og = f.root['optimizer_weights']
    names = _string_list(og.attrs, 'weight_names')
    if not names or ((len(names) - 1) % 3 != 0 and (len(names) - 1) % 2 != 0):
        raise H5FormatError('%s: /optimizer_weights does not look like Adam slots (%d entries)' % (path, len(names)))
    arrays = [np.asarray(og[n].read()) for n in names]
    it = int(np.asarray(arrays[0]).reshape(-1)[0])
    rest = arrays[1:]
    n = _adam_slot_count(names[1:], rest)
    if n is None:
        raise H5FormatError('%s: /optimizer_weights: %d arrays after the iteration counter are neither [ms, vs] nor '
                            '[ms, vs, vhats] of one set of weights' % (path, len(rest)))
    ms = [np.ascontiguousarray(a, dtype=np.float32) for a in rest[:n]]
    vs = [np.ascontiguousarray(a, dtype=np.float32) for a in rest[n:2 * n]]
    return it, ms, vs


def convert(path_h5, path_npz):
    """Keras .h5 -> the .npz layout of synthsr_amd.training.save_checkpoint (weights and BN moving statistics)"""
    sd = load_keras_weights(path_h5)
    np.savez(path_npz, **sd)
    return sd


# ---------------------------------------------------------------------- writer (the subset Keras reads back)
_KERAS_WEIGHT_ORDER = ('kernel', 'bias', 'gamma', 'beta', 'moving_mean', 'moving_variance')
_ATTR_LIMIT = 64512          # Keras' HDF5_OBJECT_HEADER_LIMIT: longer name lists are split into name0, name1, ...
_UNDEF = 0xFFFFFFFFFFFFFFFF
_INTERNAL_K = 16


def _message(mtype, data, flags=0):
    data = data + bytes(_pad8(len(data)) - len(data))
    return struct.pack('<HHB3x', mtype, len(data), flags) + data


def _object_header(messages):
    body = b''.join(messages)
    return struct.pack('<BBHII4x', 1, 0, len(messages), 1, len(body)) + body


def _padded(b):
    return b + bytes(_pad8(len(b)) - len(b))


def _dataspace_v1(shape):
    return struct.pack('<BBB5x', 1, len(shape), 0) + b''.join(struct.pack('<Q', int(s)) for s in shape)


def _datatype(dtype):
    dtype = np.dtype(dtype)
    if dtype.kind == 'S':
        return struct.pack('<BBBBI', 0x13, 0x01, 0, 0, dtype.itemsize)                      # null-padded ASCII
    if dtype.kind == 'f' and dtype.itemsize in (4, 8):
        ebits, mbits, bias = (8, 23, 127) if dtype.itemsize == 4 else (11, 52, 1023)
        return struct.pack('<BBBBIHHBBBBI', 0x11, 0x20, 8 * dtype.itemsize - 1, 0, dtype.itemsize, 0,
                           8 * dtype.itemsize, mbits, ebits, 0, mbits, bias)
    if dtype.kind in 'iu':
        return struct.pack('<BBBBIHH', 0x10, 0x08 if dtype.kind == 'i' else 0, 0, 0, dtype.itemsize, 0,
                           8 * dtype.itemsize)
    raise H5FormatError('cannot write dtype %s' % dtype)


def _attribute(name, value):
    if isinstance(value, (bytes, str)):
        value = value.encode('utf8') if isinstance(value, str) else value
        arr = np.array(value, dtype='S%d' % max(1, len(value)))
    else:
        arr = np.asarray(value)
        if arr.dtype.kind == 'U' or arr.dtype == object:
            arr = np.array([_text(v).encode('utf8') for v in arr.reshape(-1)] or [b''][:0], dtype=np.bytes_)
        if arr.dtype.kind == 'S' and arr.dtype.itemsize == 0:
            arr = arr.astype('S1')
    nm = name.encode('utf8') + b'\0'
    dt, ds = _datatype(arr.dtype), _dataspace_v1(arr.shape)
    body = struct.pack('<BxHHH', 1, len(nm), len(dt), len(ds)) + _padded(nm) + _padded(dt) + _padded(ds)
    body += arr.astype(arr.dtype.newbyteorder('<') if arr.dtype.kind in 'fiu' else arr.dtype).tobytes()
    if len(body) > 65528:
        raise H5FormatError('attribute %s too large for one object-header message' % name)
    return _message(0x0C, body)


def _string_list_attributes(name, strings):
    data = np.array([s.encode('utf8') for s in strings], dtype=np.bytes_) if strings else np.zeros((0,), 'S1')
    n = 1
    chunks = np.array_split(data, n)
    while any(c.nbytes > _ATTR_LIMIT for c in chunks):
        n += 1
        chunks = np.array_split(data, n)
    if n == 1:
        return [_attribute(name, data)]
    return [_attribute('%s%d' % (name, i), c) for i, c in enumerate(chunks)]


class _H5Writer:
    """bump allocator over one bytearray; superblock v0, symbol-table groups with a single leaf node each (the leaf
    'K' of the superblock is sized for the largest group), contiguous datasets"""

    def __init__(self, max_members):
        self.buf = bytearray(96)
        self.leaf_k = max(4, (max_members + 1) // 2)
        if self.leaf_k > 16000:
            raise H5FormatError('too many members in one group')

    def alloc(self, data):
        addr = len(self.buf)
        self.buf += _padded(bytes(data))
        return addr

    def dataset(self, arr):
        arr = np.ascontiguousarray(arr)
        le = arr.astype(arr.dtype.newbyteorder('<'))
        data = self.alloc(le.tobytes()) if arr.size else _UNDEF
        msgs = [_message(1, _dataspace_v1(arr.shape)), _message(3, _datatype(arr.dtype), 1),
                _message(5, bytes([2, 2, 2, 1, 0, 0, 0, 0]), 1),
                _message(8, struct.pack('<BBQQ', 3, 1, data, arr.nbytes))]
        return ('d', self.alloc(_object_header(msgs)))

    def group(self, members, attr_messages=()):
        """members {name: ('d', addr) | ('g', addr, btree, heap)} -> ('g', header, btree, heap)"""
        names = sorted(members, key=lambda s: s.encode('utf8'))
        heap_data, offs = bytearray(8), {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data += _padded(n.encode('utf8') + b'\0')
        data_addr = self.alloc(heap_data)
        heap = self.alloc(b'HEAP' + bytes(4) + struct.pack('<QQQ', len(heap_data), 1, data_addr))
        snod = bytearray(b'SNOD' + struct.pack('<BBH', 1, 0, len(names)))
        for n in names:
            m = members[n]
            scratch = struct.pack('<QQ', m[2], m[3]) if m[0] == 'g' else bytes(16)
            snod += struct.pack('<QQII', offs[n], m[1], 1 if m[0] == 'g' else 0, 0) + scratch
        snod += bytes(8 + 2 * self.leaf_k * 40 - len(snod))
        tree = bytearray(b'TREE' + struct.pack('<BBHQQ', 0, 0, 1 if names else 0, _UNDEF, _UNDEF) + bytes(8))
        if names:
            tree += struct.pack('<QQ', self.alloc(snod), offs[names[-1]])
        tree += bytes(24 + (2 * _INTERNAL_K + 1) * 8 + 2 * _INTERNAL_K * 8 - len(tree))
        btree = self.alloc(tree)
        hdr = self.alloc(_object_header([_message(0x11, struct.pack('<QQ', btree, heap))] + list(attr_messages)))
        return ('g', hdr, btree, heap)

    def finish(self, root, path):
        sb = SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack('<HHI', self.leaf_k, _INTERNAL_K, 0)
        sb += struct.pack('<QQQQ', 0, _UNDEF, len(self.buf), _UNDEF)
        sb += struct.pack('<QQII', 0, root[1], 1, 0) + struct.pack('<QQ', root[2], root[3])
        self.buf[:96] = sb
        with open(path, 'wb') as f:
            f.write(self.buf)


def save_keras_weights(path, state_dict, keras_version='2.3.1', backend='tensorflow'):
    """Writes {'<layer>/<weight>': array} in the layout of Keras' `model.save_weights(path)`, so that the reference's
    `model.load_weights(path, by_name=True)` (SynthSR/training.py:363, scripts/predict_command_line.py:82) can read a
    network trained here.  Layers appear in first-seen order; inside a layer the weights follow Keras' creation order
    (kernel, bias / gamma, beta, moving_mean, moving_variance), which is what by-name loading zips against."""
    layers = {}
    for key, val in state_dict.items():
        if key.startswith('optimizer/'):
            continue
        layer, w = key.rsplit('/', 1)
        layers.setdefault(layer, {})[w] = np.asarray(val)
    rank = {w: i for i, w in enumerate(_KERAS_WEIGHT_ORDER)}
    wr = _H5Writer(max([len(layers)] + [len(ws) for ws in layers.values()]))
    root_members = {}
    for layer, ws in layers.items():
        order = sorted(ws, key=lambda w: (rank.get(w, len(rank)), w))
        inner = wr.group({w + ':0': wr.dataset(ws[w].astype(np.float32)) for w in order})
        # the dataset name '<layer>/<weight>:0' is a path: a sub-group named like the layer inside the layer's group
        root_members[layer] = wr.group({layer.rsplit('/', 1)[-1]: inner} if '/' not in layer else
                                       _nest(wr, layer.split('/'), inner),
                                       _string_list_attributes('weight_names', ['%s/%s:0' % (layer, w) for w in order]))
    attrs = _string_list_attributes('layer_names', list(layers))
    attrs += [_attribute('backend', backend), _attribute('keras_version', keras_version)]
    wr.finish(wr.group(root_members, attrs), path)


def _nest(wr, parts, inner):
    for p in reversed(parts[1:]):
        inner = wr.group({p: inner})
    return {parts[0]: inner}
