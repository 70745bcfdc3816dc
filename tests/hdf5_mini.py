"""A pure-Python reader of the subset of HDF5 the reference's snapshot format
uses (test infrastructure; no h5py in the image): superblock version 0,
version-1 object headers, groups as symbol tables (version-1 B-tree nodes,
symbol-table nodes, local heap), attributes as header messages, contiguous
datasets of little-endian doubles / 32-bit integers / fixed strings. Written
after the HDF5 File Format Specification, independently of the writer's code:
it walks the structures a real HDF5 library walks.

    f = read("snapshot000.hdf5")
    f["/Header"].attrs["BoxSize"]; f["/PartType0/NeutralFractionH"][...]
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class Node:
    def __init__(self):
        self.attrs = {}
        self.members = {}
        self.data = None


def _datatype(buf):
    """(numpy dtype or ('S', size), bytes consumed is not needed)"""
    cls = buf[0] & 0x0F
    version = buf[0] >> 4
    assert version == 1, version
    bits0 = buf[1]
    size = struct.unpack_from("<I", buf, 4)[0]
    if cls == 0:   # fixed point
        assert bits0 & 1 == 0  # little endian
        signed = bool(bits0 & 0x08)
        return np.dtype("<%s%d" % ("i" if signed else "u", size))
    if cls == 1:   # floating point
        assert bits0 & 1 == 0 and size == 8
        bit_offset, precision = struct.unpack_from("<HH", buf, 8)
        eloc, esize, mloc, msize = buf[12:16]
        bias = struct.unpack_from("<I", buf, 16)[0]
        assert (bit_offset, precision, eloc, esize, mloc, msize, bias) == \
            (0, 64, 52, 11, 0, 52, 1023)
        return np.dtype("<f8")
    if cls == 3:   # string
        return ("S", size)
    raise ValueError("datatype class %d" % cls)


def _dataspace(buf):
    assert buf[0] == 1, "dataspace version"
    rank = buf[1]
    return struct.unpack_from("<%dQ" % rank, buf, 8) if rank else ()


def _pad8(n):
    return (n + 7) // 8 * 8


def _decode(dtype, dims, raw):
    if isinstance(dtype, tuple):
        return raw[:dtype[1]].split(b"\0")[0].decode()
    n = int(np.prod(dims)) if dims else 1
    a = np.frombuffer(raw[:n * dtype.itemsize], dtype=dtype)
    return a.reshape(dims).copy() if dims else a[0]


class File:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        b = self.b
        assert b[:8] == b"\x89HDF\r\n\x1a\n", "signature"
        assert b[8] == 0, "superblock version"
        assert b[13] == 8 and b[14] == 8, "offset / length sizes"
        self.leaf_k, self.internal_k = struct.unpack_from("<HH", b, 16)
        base, _, eof, _ = struct.unpack_from("<QQQQ", b, 24)
        assert base == 0 and eof == len(b), (eof, len(b))
        # root group symbol table entry
        _, header, cache = struct.unpack_from("<QQI", b, 56)
        self.root = self._object(header)

    def _messages(self, addr):
        b = self.b
        version, _, nmsg, refcount, size = struct.unpack_from("<BBHII", b,
                                                              addr)
        assert version == 1
        pos = addr + 16
        end = pos + size
        out = []
        for _ in range(nmsg):
            mtype, msize, flags = struct.unpack_from("<HHB", b, pos)
            out.append((mtype, b[pos + 8:pos + 8 + msize]))
            pos += 8 + msize
            assert pos <= end
        return out

    def _group_members(self, btree, heap):
        b = self.b
        assert b[heap:heap + 4] == b"HEAP"
        seg_size, free, seg = struct.unpack_from("<QQQ", b, heap + 8)

        def name(off):
            end = b.index(b"\0", seg + off)
            return b[seg + off:end].decode()
        members = {}

        def walk(node):
            assert b[node:node + 4] == b"TREE", "B-tree signature"
            ntype, level, used = struct.unpack_from("<BBH", b, node + 4)
            assert ntype == 0
            pos = node + 24
            for i in range(used):
                child = struct.unpack_from("<Q", b, pos + 8)[0]
                if level > 0:
                    walk(child)
                else:
                    assert b[child:child + 4] == b"SNOD"
                    nsym = struct.unpack_from("<H", b, child + 6)[0]
                    assert nsym <= 2 * self.leaf_k
                    names = []
                    for k in range(nsym):
                        off, header = struct.unpack_from("<QQ", b,
                                                         child + 8 + 40 * k)
                        names.append(name(off))
                        members[names[-1]] = header
                    assert names == sorted(names), "symbol table order"
                pos += 16
        walk(btree)
        return members

    def _object(self, addr):
        node = Node()
        layout = dtype = dims = None
        for mtype, m in self._messages(addr):
            if mtype == 0x0011:
                btree, heap = struct.unpack_from("<QQ", m, 0)
                for name, header in self._group_members(btree, heap).items():
                    node.members[name] = self._object(header)
            elif mtype == 0x000C:
                assert m[0] == 1
                nsize, tsize, ssize = struct.unpack_from("<HHH", m, 2)
                pos = 8
                name = m[pos:pos + nsize].split(b"\0")[0].decode()
                pos += _pad8(nsize)
                adt = _datatype(m[pos:pos + tsize])
                pos += _pad8(tsize)
                adims = _dataspace(m[pos:pos + ssize])
                pos += _pad8(ssize)
                node.attrs[name] = _decode(adt, adims, m[pos:])
            elif mtype == 0x0001:
                dims = _dataspace(m)
            elif mtype == 0x0003:
                dtype = _datatype(m)
            elif mtype == 0x0008:
                assert m[0] == 3 and m[1] == 1, "contiguous layout v3"
                layout = struct.unpack_from("<QQ", m, 2)
        if layout is not None:
            address, size = layout
            n = int(np.prod(dims))
            if isinstance(dtype, tuple):   # fixed-length strings
                width = dtype[1]
                assert size == n * width
                raw = self.b[address:address + size]
                node.data = [raw[i * width:(i + 1) * width].split(b"\0")[0]
                             .decode() for i in range(n)]
            else:
                assert size == n * dtype.itemsize
                node.data = np.frombuffer(self.b, dtype=dtype, count=n,
                                          offset=address).reshape(dims)
        return node

    def __getitem__(self, path):
        node = self.root
        for part in path.strip("/").split("/"):
            if part:
                node = node.members[part]
        return node


    def __contains__(self, path):
        try:
            self[path]
        except KeyError:
            return False
        return True


def read(path):
    return File(path)
