"""Hand-assembled NetCDF-4 / HDF5 files for tests/test_hdf5_cpu.py.

No HDF5 or netCDF module is importable by the tests' interpreter (h5py, netCDF4, pytables; the reference tree holds no data
files) -- files the HDF5 C library wrote are under tests/golden/hdf5/ (test_hdf5_golden_cpu.py) --, and the dialects, damage and
corner cases those do not hold are assembled here byte by byte from the
published HDF5 File Format Specification (version 3.0, sections III.A superblock, III.C object headers, III.A.1/IV.A.2 B-trees,
symbol table nodes, local heaps, IV.A.2.b dataspace, .d datatype, .i data layout, .l filter pipeline, .g link, .c link info),
in the two dialects a netCDF-4 file of five objects can come in:

* `old_style`: superblock version 0, version-1 object headers, the root group as a symbol table (B-tree version 1 + SNOD +
  local heap), `color` stored contiguously;
* `new_style`: superblock version 2, version-2 object headers ("OHDR", with a continuation block "OCHK"), compact link messages
  with creation order, `color` chunked through a two-level version-1 chunk B-tree with shuffle + deflate, big-endian floats,
  one chunk never written, one chunk stored with the deflate filter masked out.

Reader (csrc/vrm_hdf5.cpp) and these builders restate the same specification, so a misreading shared by both would not show in
test_hdf5_cpu.py alone: tests/test_hdf5_golden_cpu.py therefore hands every dialect built here to the HDF5 library's own h5dump
(where this image has it) and requires the values the tests expect -- which found two things the specification's text does not
say: B-tree and symbol-table nodes are read at their FULL size whatever they hold, and an empty local-heap free list is 1.
Checksums of version-2 structures are Jenkins' lookup3 as published (the library verifies them; the reader does not).
"""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
M = 0xFFFFFFFF


def _rot(x, k):
    return ((x << k) | (x >> (32 - k))) & M


def lookup3(data, initval=0):
    """hashlittle() of lookup3.c (Bob Jenkins, public domain), the metadata checksum of HDF5 version-2 structures."""
    n = len(data)
    a = b = c = (0xDEADBEEF + n + initval) & M
    i = 0
    while n - i > 12:
        k0, k1, k2 = struct.unpack_from("<III", data, i)
        a, b, c = (a + k0) & M, (b + k1) & M, (c + k2) & M
        a = (a - c) & M; a ^= _rot(c, 4); c = (c + b) & M
        b = (b - a) & M; b ^= _rot(a, 6); a = (a + c) & M
        c = (c - b) & M; c ^= _rot(b, 8); b = (b + a) & M
        a = (a - c) & M; a ^= _rot(c, 16); c = (c + b) & M
        b = (b - a) & M; b ^= _rot(a, 19); a = (a + c) & M
        c = (c - b) & M; c ^= _rot(b, 4); b = (b + a) & M
        i += 12
    tail = bytes(data[i:]) + b"\0" * 12
    if n - i == 0:
        return c
    k0, k1, k2 = struct.unpack_from("<III", tail, 0)
    a, b, c = (a + k0) & M, (b + k1) & M, (c + k2) & M
    c ^= b; c = (c - _rot(b, 14)) & M
    a ^= c; a = (a - _rot(c, 11)) & M
    b ^= a; b = (b - _rot(a, 25)) & M
    c ^= b; c = (c - _rot(b, 16)) & M
    a ^= c; a = (a - _rot(c, 4)) & M
    b ^= a; b = (b - _rot(a, 14)) & M
    c ^= b; c = (c - _rot(b, 24)) & M
    return c


class Image:
    """A file image that grows by aligned appends."""

    def __init__(self):
        self.b = bytearray()

    def alloc(self, n, align=8):
        while len(self.b) % align:
            self.b.append(0)
        a = len(self.b)
        self.b.extend(b"\0" * n)
        return a

    def put(self, addr, data):
        self.b[addr:addr + len(data)] = data

    def append(self, data, align=8):
        a = self.alloc(len(data), align)
        self.put(a, data)
        return a


# ---- messages ---------------------------------------------------------------------------------------------------------

def msg_dataspace_v1(dims):
    return struct.pack("<BBBB4x", 1, len(dims), 0, 0) + b"".join(struct.pack("<Q", d) for d in dims)


def msg_dataspace_v2(dims):
    return struct.pack("<BBBB", 2, len(dims), 0, 1) + b"".join(struct.pack("<Q", d) for d in dims)


def msg_datatype_f32(big_endian):
    # class 1 (floating point), version 1; bit field: byte order, padding 0, mantissa normalisation 2 (implied msb), sign at bit 31
    bits0 = (1 if big_endian else 0) | (2 << 4)
    return struct.pack("<BBBBI", 0x11, bits0, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)


def msg_layout_contiguous(addr, size, version=3):
    return struct.pack("<BBQQ", version, 1, addr, size)


def msg_layout_chunked_v3(btree, chunk):
    return struct.pack("<BBBQ", 3, 2, len(chunk) + 1, btree) + b"".join(struct.pack("<I", c) for c in chunk) + struct.pack("<I", 4)


def msg_filters_v2(ids):
    out = struct.pack("<BB", 2, len(ids))
    for fid in ids:
        values = {1: [6], 2: [4], 3: []}[fid]      # deflate level, shuffle element size, fletcher32: none
        out += struct.pack("<HHH", fid, 0, len(values)) + b"".join(struct.pack("<I", v) for v in values)
    return out


def msg_filters_v1(ids):
    out = struct.pack("<BB6x", 1, len(ids))
    for fid in ids:
        values = {1: [6], 2: [4], 3: []}[fid]
        name = {1: b"deflate\0", 2: b"shuffle\0", 3: b"fletcher32\0\0\0\0\0\0"}[fid]
        out += struct.pack("<HHHH", fid, len(name), 0, len(values)) + name + b"".join(struct.pack("<I", v) for v in values)
        if len(values) % 2:
            out += b"\0" * 4
    return out


def msg_symbol_table(btree, heap):
    return struct.pack("<QQ", btree, heap)


def msg_link(name, addr, order):
    nb = name.encode()
    return struct.pack("<BB", 1, 0x04) + struct.pack("<Q", order) + struct.pack("<B", len(nb)) + nb + struct.pack("<Q", addr)


def msg_link_info():
    return struct.pack("<BBQQQQ", 0, 0x03, 4, UNDEF, UNDEF, UNDEF)   # creation order tracked + indexed, compact storage


def _attribute_v1(name, dt, ds, value):
    nb = name.encode() + b"\0"

    def pad8(b):
        return b + b"\0" * (-len(b) % 8)
    return struct.pack("<BxHHH", 1, len(nb), len(dt), len(ds)) + pad8(nb) + pad8(dt) + pad8(ds) + value


def msg_attribute_v1(name, text):
    """A scalar fixed-length string attribute (the kind netcdf-c hangs on everything); the reader ignores attributes."""
    tb = text.encode()
    dt = struct.pack("<BBBBI", 0x13, 0, 0, 0, len(tb))          # class 3 (string), null-terminated ASCII
    return _attribute_v1(name, dt, msg_dataspace_v1([]), tb)


def msg_attribute_i32(name, values, scalar):
    """A little-endian signed 32-bit attribute: a scalar or a vector (netcdf-c's _Netcdf4Dimid / _Netcdf4Coordinates)."""
    dt = struct.pack("<BBBBI", 0x10, 0x08, 0, 0, 4) + struct.pack("<HH", 0, 32)   # class 0, signed; bit offset 0, precision 32
    ds = msg_dataspace_v1([]) if scalar else msg_dataspace_v1([len(values)])
    return _attribute_v1(name, dt, ds, b"".join(struct.pack("<i", v) for v in values))


# ---- object headers -----------------------------------------------------------------------------------------------------

def object_header_v1(img, messages):
    body = b""
    for mtype, data in messages:
        data = data + b"\0" * (-len(data) % 8)
        body += struct.pack("<HHB3x", mtype, len(data), 0) + data
    return img.append(struct.pack("<BxHII4x", 1, len(messages), 1, len(body)) + body)


def object_header_v2(img, messages, continuation_after=None):
    """Version-2 header with creation-order tracking and timestamps; messages beyond `continuation_after` go to an OCHK block."""
    def enc(msgs):
        out = b""
        for i, (mtype, data) in enumerate(msgs):
            out += struct.pack("<BHBH", mtype, len(data), 0, i) + data
        return out
    first = messages if continuation_after is None else messages[:continuation_after]
    rest = [] if continuation_after is None else messages[continuation_after:]
    cont_addr = None
    if rest:
        blk = b"OCHK" + enc(rest) + struct.pack("<BHBH", 0, 3, 0, 99) + b"\0\0\0"   # + a NIL message as padding
        blk += struct.pack("<I", lookup3(blk))
        cont_addr = img.append(blk)
        first = first + [(0x10, struct.pack("<QQ", cont_addr, len(blk)))]
    body = enc(first) + b"\0\0\0"                                                      # a 3-byte gap: too short for a message
    flags = 0x04 | 0x20 | 0x01                                                          # order tracked, times stored, 2-byte chunk size
    head = b"OHDR" + struct.pack("<BB", 2, flags) + struct.pack("<IIII", 1, 2, 3, 4) + struct.pack("<H", len(body))
    blk = head + body
    blk += struct.pack("<I", lookup3(blk))
    return img.append(blk)


# ---- files ----------------------------------------------------------------------------------------------------------------

def dimension_scale_messages(n, v2):
    ds = msg_dataspace_v2([n]) if v2 else msg_dataspace_v1([n])
    return [(0x01, ds), (0x03, msg_datatype_f32(True)), (0x08, msg_layout_contiguous(UNDEF, 0)),
            (0x0C, msg_attribute_v1("CLASS", "DIMENSION_SCALE\0"))]


def netcdf4_classic_model(color):
    """The file csrc/vrm_hdf5.cpp's writer (vrm_netcdf_write_color_nc4) makes for color(4, Z, Y, X), assembled here
    independently: the `old_style` dialect with netcdf-c's attributes (CLASS, NAME, _Netcdf4Dimid on the dimension scales,
    _Netcdf4Coordinates on the variable), full-size symbol and B-tree nodes, and the root entry's scratch filled in."""
    color = np.ascontiguousarray(color, dtype="<f4")
    _, Z, Y, X = color.shape
    img = Image()
    sb = img.alloc(96)
    data_addr = img.append(color.tobytes())
    fill = (0x05, struct.pack("<BBBB", 2, 2, 2, 0))

    def scale(i, n):
        name = "This is a netCDF dimension but not a netCDF variable.%10d" % n
        return object_header_v1(img, [(0x01, msg_dataspace_v1([n])), (0x03, msg_datatype_f32(True)), fill,
                                      (0x08, msg_layout_contiguous(UNDEF, 4 * n)), (0x0C, msg_attribute_v1("CLASS", "DIMENSION_SCALE\0")),
                                      (0x0C, msg_attribute_v1("NAME", name + "\0")), (0x0C, msg_attribute_i32("_Netcdf4Dimid", [i], True))])
    objs = {"c": scale(0, 4), "z": scale(1, Z), "y": scale(2, Y), "x": scale(3, X)}
    objs["color"] = object_header_v1(img, [(0x01, msg_dataspace_v1(color.shape)), (0x03, msg_datatype_f32(False)), fill,
                                           (0x08, msg_layout_contiguous(data_addr, color.nbytes)),
                                           (0x0C, msg_attribute_i32("_Netcdf4Coordinates", [0, 1, 2, 3], False))])
    names = sorted(objs)
    heap_data = bytearray(b"\0" * 8)
    name_off = {}
    for n in names:
        name_off[n] = len(heap_data)
        heap_data += n.encode() + b"\0"
        heap_data += b"\0" * (-len(heap_data) % 8)
    heap_data_addr = img.append(bytes(heap_data))
    heap = img.append(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, heap_data_addr))  # free list empty: libhdf5 writes 1 (H5HL_FREE_NULL), found with the real library (tests/test_hdf5_golden_cpu.py)
    leaf_k, internal_k = 4, 16
    snod = b"SNOD" + struct.pack("<BxH", 1, len(names))
    for n in names:
        snod += struct.pack("<QQII16x", name_off[n], objs[n], 0, 0)
    snod += b"\0" * (40 * (2 * leaf_k - len(names)))
    snod_addr = img.append(snod)
    tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod_addr, name_off[names[-1]])
    tree += b"\0" * (16 * (2 * internal_k - 1))
    tree_addr = img.append(tree)
    root = object_header_v1(img, [(0x11, msg_symbol_table(tree_addr, heap)),
                                  (0x0C, msg_attribute_v1("_NCProperties", "version=2,volume_renderer_amd=3\0"))])
    eof = len(img.b)
    sblock = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", leaf_k, internal_k, 0)
    sblock += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sblock += struct.pack("<QQIIQQ", 0, root, 1, 0, tree_addr, heap)
    assert len(sblock) == 96
    img.put(sb, sblock)
    return bytes(img.b)


def old_style(color, dims_extra=None):
    """color: float32 (4, Z, Y, X).  Superblock v0, symbol-table root group, contiguous little-endian dataset."""
    color = np.ascontiguousarray(color, dtype="<f4")
    _, Z, Y, X = color.shape
    img = Image()
    sb = img.alloc(8 + 8 + 4 + 4 + 4 * 8 + 40)      # signature + versions/sizes + K + flags + 4 addresses + root entry
    data_addr = img.append(color.tobytes())
    objs = {"c": object_header_v1(img, dimension_scale_messages(color.shape[0], False)),
            "z": object_header_v1(img, dimension_scale_messages((dims_extra or {}).get("z", Z), False)),
            "y": object_header_v1(img, dimension_scale_messages((dims_extra or {}).get("y", Y), False)),
            "x": object_header_v1(img, dimension_scale_messages((dims_extra or {}).get("x", X), False)),
            "color": object_header_v1(img, [(0x01, msg_dataspace_v1(color.shape)), (0x03, msg_datatype_f32(False)),
                                            (0x05, struct.pack("<BBBB", 2, 2, 2, 0)),          # fill value message (ignored)
                                            (0x08, msg_layout_contiguous(data_addr, color.nbytes)),
                                            (0x0C, msg_attribute_v1("_Netcdf4Dimid", "0"))])}
    names = sorted(objs)                              # symbol table entries are sorted by name
    heap_data = bytearray(b"\0" * 8)                  # offset 0: the empty name of the B-tree's first key
    name_off = {}
    for n in names:
        name_off[n] = len(heap_data)
        heap_data += n.encode() + b"\0"
        heap_data += b"\0" * (-len(heap_data) % 8)
    heap_data_addr = img.append(bytes(heap_data))
    heap = img.append(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, heap_data_addr))  # free list empty: libhdf5 writes 1 (H5HL_FREE_NULL), found with the real library (tests/test_hdf5_golden_cpu.py)
    snod = b"SNOD" + struct.pack("<BxH", 1, len(names))
    for n in names:
        snod += struct.pack("<QQII16x", name_off[n], objs[n], 0, 0)
    snod += b"\0" * ((2 * 4 - len(names)) * 40)      # (nodes are stored at their full size -- 2K entries, 2K children and 2K+1
    snod_addr = img.append(snod)                     # keys -- whatever they hold: libhdf5 reads the whole node)
    tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod_addr, name_off[names[-1]])
    tree += b"\0" * ((2 * 16 - 1) * 16)
    tree_addr = img.append(tree)
    root = object_header_v1(img, [(0x11, msg_symbol_table(tree_addr, heap)), (0x0C, msg_attribute_v1("_NCProperties", "version=2"))])
    eof = len(img.b)
    sblock = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", 4, 16, 0)
    sblock += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sblock += struct.pack("<QQII16x", 0, root, 1, 0)   # root symbol table entry (cache type 1: the scratch holds B-tree / heap addresses)
    assert len(sblock) == 96
    img.put(sb, sblock)
    return bytes(img.b)


def new_style(color, chunk=(2, 3, 4, 5), skip_chunk=1, unfiltered_chunk=2, filters=(2, 1), v1_pipeline=False, superblock=2):
    """color: float32 (4, Z, Y, X).  Superblock v2/v3, OHDR object headers, compact links, chunked + filtered big-endian dataset
    behind a two-level chunk B-tree.  Chunk number `skip_chunk` (in row-major chunk order) is never written (reads as 0), chunk
    number `unfiltered_chunk` is stored with the deflate filter masked out.  Returns (file bytes, the array a reader must see)."""
    color = np.asarray(color, dtype=np.float32)
    C, Z, Y, X = color.shape
    img = Image()
    sb = img.alloc(8 + 4 + 4 * 8 + 4)
    expect = color.copy()
    grid = [range(0, d, c) for d, c in zip(color.shape, chunk)]
    leaves = []
    k = 0
    deflate_index = list(filters).index(1) if 1 in filters else None
    for a in grid[0]:
        for b in grid[1]:
            for c in grid[2]:
                for d in grid[3]:
                    origin = (a, b, c, d)
                    if k == skip_chunk:
                        expect[a:a + chunk[0], b:b + chunk[1], c:c + chunk[2], d:d + chunk[3]] = 0
                        k += 1
                        continue
                    full = np.full(chunk, 7.5, dtype=">f4")      # edge chunks are stored whole; the overhang holds junk
                    part = color[a:a + chunk[0], b:b + chunk[1], c:c + chunk[2], d:d + chunk[3]]
                    full[:part.shape[0], :part.shape[1], :part.shape[2], :part.shape[3]] = part
                    raw = full.tobytes()
                    mask = 0
                    for i, fid in enumerate(filters):
                        if fid == 1 and k == unfiltered_chunk:
                            mask |= 1 << i
                            continue
                        if fid == 2:
                            arr = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 4)
                            raw = arr.T.tobytes()
                        elif fid == 1:
                            raw = zlib.compress(raw, 6)
                        elif fid == 3:
                            raw = raw + b"\xde\xad\xbe\xef"          # the checksum (not verified by the reader)
                    addr = img.append(raw)
                    leaves.append((len(raw), mask, origin, addr))
                    k += 1
    assert deflate_index is None or unfiltered_chunk is None or unfiltered_chunk < k

    def key(size, mask, origin):
        return struct.pack("<II", size, mask) + b"".join(struct.pack("<Q", o) for o in origin) + struct.pack("<Q", 0)

    def node(level, entries, last_origin):
        body = b""
        for size, mask, origin, child in entries:
            body += key(size, mask, origin) + struct.pack("<Q", child)
        body += key(0, 0, last_origin)
        body += b"\0" * (24 + 64 * 8 + 65 * 48 - 24 - len(body))   # stored at full size (K = 32: 64 children, 65 keys): libhdf5 reads all of it
        return img.append(b"TREE" + struct.pack("<BBHQQ", 1, level, len(entries), UNDEF, UNDEF) + body)
    end_origin = tuple(color.shape)
    half = max(1, len(leaves) // 2)
    groups = [leaves[:half], leaves[half:]] if len(leaves) > 1 else [leaves]
    children = []
    for g in groups:
        if g:
            children.append((g[0][0], g[0][1], g[0][2], node(0, g, end_origin)))
    btree = node(1, children, end_origin)
    pipeline = msg_filters_v1(filters) if v1_pipeline else msg_filters_v2(filters)
    msgs = [(0x01, msg_dataspace_v2(color.shape)), (0x03, msg_datatype_f32(True))]
    if filters:
        msgs.append((0x0B, pipeline))
    msgs += [(0x08, msg_layout_chunked_v3(btree, chunk)), (0x0C, msg_attribute_v1("DIMENSION_LIST", "refs"))]
    objs = [("c", object_header_v2(img, dimension_scale_messages(C, True))), ("z", object_header_v2(img, dimension_scale_messages(Z, True))),
            ("y", object_header_v2(img, dimension_scale_messages(Y, True))), ("x", object_header_v2(img, dimension_scale_messages(X, True))),
            ("color", object_header_v2(img, msgs, continuation_after=2))]
    root_msgs = [(0x02, msg_link_info()), (0x0A, struct.pack("<BB", 0, 0))] + [(0x06, msg_link(n, a, i)) for i, (n, a) in enumerate(objs)]
    root = object_header_v2(img, root_msgs, continuation_after=4)      # three of the five links live in the continuation block
    eof = len(img.b)
    sblock = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBB", superblock, 8, 8, 0) + struct.pack("<QQQQ", 0, UNDEF, eof, root)
    sblock += struct.pack("<I", lookup3(sblock))
    img.put(sb, sblock)
    return bytes(img.b), expect


def single_chunk_v4(color, deflate=True):
    """Layout message version 4 with the 'single chunk' index (HDF5 1.10 'latest' format): the whole dataset is one chunk."""
    color = np.ascontiguousarray(color, dtype="<f4")
    img = Image()
    sb = img.alloc(8 + 4 + 4 * 8 + 4)
    raw = color.tobytes()
    if deflate:
        raw = zlib.compress(raw, 6)
    addr = img.append(raw)
    layout = struct.pack("<BBBBB", 4, 2, 0x02 if deflate else 0x00, 5, 4) + b"".join(struct.pack("<I", d) for d in color.shape) + struct.pack("<I", 4)
    layout += struct.pack("<B", 1)
    if deflate:
        layout += struct.pack("<QI", len(raw), 0)
    layout += struct.pack("<Q", addr)
    msgs = [(0x01, msg_dataspace_v2(color.shape)), (0x03, msg_datatype_f32(False))]
    if deflate:
        msgs.append((0x0B, msg_filters_v2([1])))
    msgs.append((0x08, layout))
    col = object_header_v2(img, msgs)
    root = object_header_v2(img, [(0x02, msg_link_info()), (0x06, msg_link("color", col, 0))])
    eof = len(img.b)
    sblock = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBB", 3, 8, 8, 0) + struct.pack("<QQQQ", 0, UNDEF, eof, root)
    sblock += struct.pack("<I", lookup3(sblock))
    img.put(sb, sblock)
    return bytes(img.b)
