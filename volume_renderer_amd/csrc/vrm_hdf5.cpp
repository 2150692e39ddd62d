// vrm_hdf5.cpp -- the NetCDF-4 container of the reference's own volume files.
//
// `netcdf::create` / `netcdf::open` in loaders/netcdf.rs:10,35 and `Dataset(..., format='NETCDF4_CLASSIC')` in
// scripts/diffdvr.py:278 produce NetCDF-4 files, i.e. HDF5 containers: the variable `color(c,z,y,x)` is an HDF5 dataset
// named "color" in the root group, the dimensions c, z, y, x are (dimension-scale) datasets beside it.  netcdf-c and the
// HDF5 library are dependencies of the reference that are not in its tree and that this library does not link, so this is
// a reader for the subset of the published HDF5 File Format Specification (version 3.0) that such files use, restated
// from the specification:
//   * superblock versions 0, 1 (symbol-table root) and 2, 3 (root object header address);
//   * object headers version 1 and version 2 ("OHDR"), with continuation blocks ("OCHK" for version 2);
//   * groups as a symbol table (B-tree version 1 "TREE" of "SNOD" nodes + local "HEAP"), as compact link messages
//     (what netcdf-c's creation-order tracking produces for a handful of objects) and as dense link storage (more than
//     8 objects: fractal heap "FRHP" + version-2 B-tree name index, up to about a thousand links);
//   * the dataset's dataspace (versions 1, 2), datatype (IEEE f32; also f64 and integers of 1-8 bytes, converted to f32 the way
//     netcdf-c's nc_get_vara_float does for the reference's get_values::<f32>; little or big endian), fill value, and data layout
//     message version 3 -- compact, contiguous, or chunked through a version-1 chunk B-tree -- and version 4: compact,
//     contiguous, single-chunk, implicit and fixed-array chunk indexes (the indexes of datasets with unlimited dimensions,
//     extensible array and version-2 B-tree, are recognised and refused);
//   * the filter pipeline with deflate (zlib), shuffle and fletcher32 (checksum not verified), the filters netCDF-4 offers.
// PINNED against the HDF5 library itself (round 3): tests/golden/hdf5/lib_*.nc were written by libhdf5 1.10.6 through the
// calls netcdf-c makes (tools/make_hdf5_golden.c) in every one of these dialects and are read back bit for bit
// (tests/test_hdf5_golden_cpu.py); the byte-built files of tests/hdf5_builder.py are read by the library's own h5dump with
// the values the tests expect.  What stays unpinned is netcdf-c's layer above HDF5 (absent here).
#include <zlib.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "vrm_internal.h"

int vrm_hdf5_read_color(FILE *f, const char *path, uint32_t *x, uint32_t *y, uint32_t *z, float *out, size_t capacity_floats);

namespace {

const uint64_t UNDEF = ~0ull;

struct Fail {
    std::string msg;
};

struct H5 {
    FILE *f;
    const char *path;
    uint64_t base = 0, file_size = 0;
    int so = 8, sl = 8; // size of offsets / lengths

    [[noreturn]] void fail(const char *fmt, ...) __attribute__((format(printf, 2, 3)))
    {
        char buf[400];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        throw Fail{std::string(path) + ": " + buf};
    }
    void read(uint64_t addr, void *dst, size_t n)
    {
        if (addr == UNDEF || addr + base + n > file_size || addr + base + n < addr) fail("read of %zu bytes at %llu beyond the end of the file", n, (unsigned long long)addr);
        if (fseeko(f, (off_t)(addr + base), SEEK_SET) != 0 || fread(dst, 1, n, f) != n) fail("cannot read %zu bytes at %llu", n, (unsigned long long)addr);
    }
    std::vector<uint8_t> block(uint64_t addr, size_t n)
    {
        if (n > ((size_t)1 << 28)) fail("metadata block of %zu bytes is not plausible", n);
        std::vector<uint8_t> b(n);
        read(addr, b.data(), n);
        return b;
    }
};

// little-endian field reader over a metadata block
struct Cur {
    H5 &h;
    const uint8_t *p, *end;
    Cur(H5 &h_, const uint8_t *b, size_t n) : h(h_), p(b), end(b + n) {}
    void need(size_t n) { if ((size_t)(end - p) < n) h.fail("truncated metadata"); }
    uint64_t u(int n)
    {
        need((size_t)n);
        uint64_t v = 0;
        for (int i = 0; i < n; ++i) v |= (uint64_t)p[i] << (8 * i);
        p += n;
        return v;
    }
    uint64_t off() { const uint64_t v = u(h.so); return (h.so < 8 && v == (1ull << (8 * h.so)) - 1) ? UNDEF : v; }
    uint64_t len() { return u(h.sl); }
    void skip(size_t n) { need(n); p += n; }
    size_t left() const { return (size_t)(end - p); }
};

struct Message {
    uint16_t type;
    uint8_t flags;
    std::vector<uint8_t> data;
};

// every message of an object header (version 1 or 2), continuation blocks followed
std::vector<Message> object_header(H5 &h, uint64_t addr)
{
    std::vector<Message> out;
    uint8_t sig[4];
    h.read(addr, sig, 4);
    struct Span { uint64_t addr, size; };
    std::vector<Span> todo;
    bool v2 = memcmp(sig, "OHDR", 4) == 0;
    bool track_order = false;
    if (v2) {
        if (addr >= h.file_size) h.fail("object header beyond the end of the file");
        auto b = h.block(addr, (size_t)std::min<uint64_t>(64, h.file_size - addr)); // (a short header may end the file)
        Cur c(h, b.data(), b.size());
        c.skip(4);
        if (c.u(1) != 2) h.fail("object header version is not 2");
        const uint8_t flags = (uint8_t)c.u(1);
        if (flags & 0x20) c.skip(16);
        if (flags & 0x10) c.skip(4);
        const uint64_t sz = c.u(1 << (flags & 3));
        track_order = flags & 0x04;
        todo.push_back({addr + (uint64_t)(c.p - b.data()), sz});
    } else {
        auto b = h.block(addr, 16);
        Cur c(h, b.data(), b.size());
        if (c.u(1) != 1) h.fail("object header at %llu is neither version 1 nor version 2", (unsigned long long)addr);
        c.skip(1);
        c.u(2); // number of messages: the sizes bound the walk
        c.skip(4);
        const uint64_t sz = c.u(4);
        todo.push_back({addr + 16, sz});
    }
    for (size_t t = 0; t < todo.size(); ++t) {
        if (todo.size() > 4096) h.fail("object header continuation chain is not plausible");
        uint64_t a = todo[t].addr, n = todo[t].size;
        if (t > 0 && v2) { // "OCHK" + messages + checksum
            uint8_t s2[4];
            h.read(a, s2, 4);
            if (memcmp(s2, "OCHK", 4) != 0) h.fail("bad continuation block signature");
            a += 4;
            n -= 8;
        }
        auto b = h.block(a, (size_t)n);
        Cur c(h, b.data(), b.size());
        const size_t hdr = v2 ? (size_t)(4 + (track_order ? 2 : 0)) : 8;
        while (c.left() >= hdr) {
            Message m;
            uint64_t sz;
            if (v2) {
                m.type = (uint16_t)c.u(1);
                sz = c.u(2);
                m.flags = (uint8_t)c.u(1);
                if (track_order) c.skip(2);
            } else {
                m.type = (uint16_t)c.u(2);
                sz = c.u(2);
                m.flags = (uint8_t)c.u(1);
                c.skip(3);
            }
            if (sz > c.left()) { if (v2) break; h.fail("object header message overruns its block"); } // (v2: the gap before the checksum)
            m.data.assign(c.p, c.p + sz);
            c.skip((size_t)sz);
            if (m.type == 0x0010) { // continuation
                Cur cc(h, m.data.data(), m.data.size());
                const uint64_t ca = cc.off(), cl = cc.len();
                todo.push_back({ca, cl});
            } else if (m.type != 0) {
                out.push_back(std::move(m));
            }
        }
    }
    return out;
}

typedef std::vector<std::pair<std::string, uint64_t>> Links;

// one link message (in an object header, or as an object of a dense group's fractal heap)
void parse_link(H5 &h, const uint8_t *data, size_t n, Links &links)
{
    Cur c(h, data, n);
    if (c.u(1) != 1) h.fail("link message version is not 1");
    const uint8_t fl = (uint8_t)c.u(1);
    uint8_t ltype = 0;
    if (fl & 0x08) ltype = (uint8_t)c.u(1);
    if (fl & 0x04) c.skip(8);
    if (fl & 0x10) c.skip(1);
    const uint64_t nlen = c.u(1 << (fl & 3));
    c.need((size_t)nlen);
    std::string name(reinterpret_cast<const char *>(c.p), (size_t)nlen);
    c.skip((size_t)nlen);
    if (ltype == 0) links.emplace_back(name, c.off()); // hard link; soft / external links are not followed
}

int bits_of(uint64_t v) { int b = 0; while (v >>= 1) ++b; return b; } // floor(log2 v)

// Dense link storage (a group of more than 8 objects, or one that has held that many): the links are objects of a fractal heap
// "FRHP", found through the records of the name index, a version-2 B-tree "BTHD" of type 5 (hash of the name + heap ID).
// Read: heaps without I/O filters whose root is a direct block or an indirect block of direct blocks (group heaps start with
// 512-byte blocks, four to a row, up to 64 KB: half a megabyte of links), managed objects only; B-trees of depth 0 and 1.
void dense_links(H5 &h, uint64_t heap_addr, uint64_t index_addr, Links &links)
{
    auto hb = h.block(heap_addr, (size_t)(4 + 1 + 2 + 2 + 1 + 4 + h.sl + h.so + h.sl + h.so + 8 * h.sl + 2 + 2 * h.sl + 2 + 2 + h.so + 2));
    if (memcmp(hb.data(), "FRHP", 4) != 0) h.fail("bad fractal heap signature");
    Cur c(h, hb.data() + 4, hb.size() - 4);
    if (c.u(1) != 0) h.fail("fractal heap version");
    const uint64_t id_len = c.u(2), filter_len = c.u(2);
    c.u(1);                  // flags (bit 1: direct blocks are checksummed -- the checksum trails the block, nothing moves)
    const uint64_t max_managed = c.u(4);
    c.len(); c.off();        // huge objects
    c.len(); c.off();        // free space
    c.len(); c.len(); c.len(); c.len(); // managed space, allocated, iterator offset, number of managed objects
    c.len(); c.len(); c.len(); c.len(); // huge / tiny sizes and counts
    const uint64_t width = c.u(2), start = c.len(), max_direct = c.len(), heap_bits = c.u(2);
    c.u(2);
    const uint64_t root = c.off(), root_rows = c.u(2);
    if (filter_len != 0) h.fail("fractal heap with I/O filters is not read");
    if (width == 0 || width > 1024 || start < 16 || (start & (start - 1)) || max_direct < start || (max_direct & (max_direct - 1)) || max_direct > ((uint64_t)1 << 26) || heap_bits == 0 || heap_bits > 64)
        h.fail("fractal heap parameters are not plausible");
    const size_t off_size = (size_t)((heap_bits + 7) / 8);
    const size_t len_size = (size_t)std::min<int>((bits_of(max_direct) + 7) / 8, bits_of(max_managed ? max_managed : 1) / 8 + 1);
    if (1 + off_size + len_size > id_len) h.fail("fractal heap IDs of %llu bytes", (unsigned long long)id_len);
    // the direct blocks: (offset in the heap's address space, size, file address)
    struct Block { uint64_t off, size, addr; };
    std::vector<Block> blocks;
    if (root == UNDEF) return; // an empty heap
    if (root_rows == 0) {
        blocks.push_back({0, start, root});
    } else {
        const uint64_t max_direct_rows = (uint64_t)(bits_of(max_direct) - bits_of(start)) + 2;
        if (root_rows > max_direct_rows) h.fail("the root group holds more links than this reader follows (fractal heap with nested indirect blocks)");
        auto ib = h.block(root, (size_t)(4 + 1 + h.so + off_size + root_rows * width * h.so));
        if (memcmp(ib.data(), "FHIB", 4) != 0) h.fail("bad fractal heap indirect block signature");
        Cur ic(h, ib.data() + 5 + h.so + off_size, ib.size() - 5 - (size_t)h.so - off_size);
        uint64_t off = 0;
        for (uint64_t r = 0; r < root_rows; ++r) {
            const uint64_t size = r < 2 ? start : start << (r - 1);
            for (uint64_t k = 0; k < width; ++k, off += size) {
                const uint64_t a = ic.off();
                if (a != UNDEF) blocks.push_back({off, size, a});
            }
        }
    }
    auto heap_object = [&](const uint8_t *id) {
        if ((id[0] & 0x30) != 0) h.fail("a link is stored as a huge or tiny heap object: not read");
        uint64_t off = 0, len = 0;
        for (size_t i = 0; i < off_size; ++i) off |= (uint64_t)id[1 + i] << (8 * i);
        for (size_t i = 0; i < len_size; ++i) len |= (uint64_t)id[1 + off_size + i] << (8 * i);
        for (const Block &b : blocks)
            if (off >= b.off && off - b.off < b.size) {
                if (len == 0 || len > b.size - (off - b.off)) break;
                auto data = h.block(b.addr + (off - b.off), (size_t)len);
                parse_link(h, data.data(), data.size(), links);
                return;
            }
        h.fail("a link's heap ID points outside the heap's blocks");
    };
    // the name index
    auto tb = h.block(index_addr, (size_t)(4 + 1 + 1 + 4 + 2 + 2 + 1 + 1 + h.so + 2 + h.sl));
    if (memcmp(tb.data(), "BTHD", 4) != 0) h.fail("bad version-2 B-tree header signature");
    Cur tc(h, tb.data() + 4, tb.size() - 4);
    if (tc.u(1) != 0) h.fail("version-2 B-tree header version");
    const uint64_t type = tc.u(1), node_size = tc.u(4), rec_size = tc.u(2), depth = tc.u(2);
    tc.u(2);
    const uint64_t root_node = tc.off(), root_nrec = tc.u(2);
    if (type != 5 || rec_size != 4 + id_len || node_size < 32 || node_size > ((uint64_t)1 << 24)) h.fail("the root group's name index is not a type-5 B-tree of hash + heap ID records");
    if (depth > 1) h.fail("the root group holds more links than this reader follows (name index of depth %llu)", (unsigned long long)depth);
    if (root_node == UNDEF) return;
    auto leaf = [&](uint64_t addr, uint64_t nrec) {
        if (nrec * rec_size + 10 > node_size) h.fail("version-2 B-tree leaf of %llu records", (unsigned long long)nrec);
        auto nb = h.block(addr, (size_t)(6 + nrec * rec_size));
        if (memcmp(nb.data(), "BTLF", 4) != 0) h.fail("bad version-2 B-tree leaf signature");
        for (uint64_t i = 0; i < nrec; ++i) heap_object(nb.data() + 6 + i * rec_size + 4);
    };
    if (depth == 0) {
        leaf(root_node, root_nrec);
    } else {
        const uint64_t max_leaf = (node_size - 10) / rec_size;
        const size_t nrec_size = (size_t)(bits_of(max_leaf ? max_leaf : 1) / 8 + 1);
        if (root_nrec * rec_size + (root_nrec + 1) * (h.so + nrec_size) + 10 > node_size) h.fail("version-2 B-tree internal node of %llu records", (unsigned long long)root_nrec);
        auto nb = h.block(root_node, (size_t)(6 + root_nrec * rec_size + (root_nrec + 1) * (h.so + nrec_size)));
        if (memcmp(nb.data(), "BTIN", 4) != 0) h.fail("bad version-2 B-tree internal node signature");
        for (uint64_t i = 0; i < root_nrec; ++i) heap_object(nb.data() + 6 + i * rec_size + 4);
        Cur pc(h, nb.data() + 6 + root_nrec * rec_size, (size_t)((root_nrec + 1) * (h.so + nrec_size)));
        for (uint64_t i = 0; i <= root_nrec; ++i) {
            const uint64_t child = pc.off(), n = pc.u((int)nrec_size);
            leaf(child, n);
        }
    }
}

// name -> object header address of every link of a group
void group_links(H5 &h, const std::vector<Message> &msgs, Links &links)
{
    for (const Message &m : msgs) {
        if (m.type == 0x0011) { // symbol table: B-tree v1 + local heap
            Cur c(h, m.data.data(), m.data.size());
            const uint64_t btree = c.off(), heap = c.off();
            auto hb = h.block(heap, (size_t)(8 + 2 * h.sl + h.so));
            if (memcmp(hb.data(), "HEAP", 4) != 0) h.fail("bad local heap signature");
            Cur hc(h, hb.data() + 8, hb.size() - 8);
            const uint64_t hsize = hc.len();
            hc.len();
            const uint64_t hdata = hc.off();
            auto names = h.block(hdata, (size_t)hsize);
            std::vector<uint64_t> nodes{btree};
            size_t visited = 0;
            while (!nodes.empty()) {
                if (++visited > 100000) h.fail("group B-tree is not plausible");
                const uint64_t a = nodes.back();
                nodes.pop_back();
                uint8_t sig[4];
                h.read(a, sig, 4);
                if (memcmp(sig, "TREE", 4) == 0) {
                    auto nb = h.block(a, (size_t)(8 + 2 * h.so));
                    Cur nc(h, nb.data() + 4, nb.size() - 4);
                    if (nc.u(1) != 0) h.fail("group B-tree node has the wrong type");
                    nc.u(1);
                    const uint64_t used = nc.u(2);
                    auto body = h.block(a + 8 + 2 * (uint64_t)h.so, (size_t)((used + 1) * h.sl + used * h.so));
                    Cur bc(h, body.data(), body.size());
                    for (uint64_t i = 0; i < used; ++i) {
                        bc.len(); // key
                        nodes.push_back(bc.off());
                    }
                } else if (memcmp(sig, "SNOD", 4) == 0) {
                    auto nb = h.block(a, 8);
                    Cur nc(h, nb.data() + 4, 4);
                    nc.skip(2);
                    const uint64_t n = nc.u(2);
                    const size_t esz = (size_t)(2 * h.so + 24);
                    auto body = h.block(a + 8, (size_t)n * esz);
                    for (uint64_t i = 0; i < n; ++i) {
                        Cur ec(h, body.data() + i * esz, esz);
                        const uint64_t name_off = ec.off(), ohdr = ec.off();
                        if (name_off >= names.size()) h.fail("link name outside the local heap");
                        const char *s = reinterpret_cast<const char *>(names.data() + name_off);
                        links.emplace_back(std::string(s, strnlen(s, names.size() - (size_t)name_off)), ohdr);
                    }
                } else {
                    h.fail("unexpected node in a group B-tree");
                }
            }
        } else if (m.type == 0x0006) { // link message
            parse_link(h, m.data.data(), m.data.size(), links);
        } else if (m.type == 0x0002) { // link info: dense storage?
            Cur c(h, m.data.data(), m.data.size());
            c.u(1);
            const uint8_t fl = (uint8_t)c.u(1);
            if (fl & 1) c.skip(8);
            const uint64_t heap = c.off(), index = c.off();
            if (heap != UNDEF) dense_links(h, heap, index, links);
        }
    }
}

struct Dataset {
    std::vector<uint64_t> dims;
    bool big_endian = false;
    // the stored element: IEEE float of 4 or 8 bytes, or an integer of 1, 2, 4 or 8 bytes; whatever it is, the variable is handed
    // over as f32 the way netcdf-c's nc_get_vara_float -- what the reference's get_values::<f32> (netcdf.rs:45) calls -- converts it
    size_t esize = 4;
    bool is_float = true, is_signed = true;
    uint64_t chunk_elem = 0;   // the element size the layout message states (checked against the datatype once both are known)
    uint32_t shuffle_size = 0; // the shuffle filter's own element size
    int layout = -1; // 0 compact, 1 contiguous, 2 chunked (B-tree v1), 3 single chunk, 4 implicit chunk index, 5 fixed-array chunk index
    uint64_t addr = UNDEF, size = 0;
    std::vector<uint8_t> compact; // layout 0: the data itself, from the layout message
    bool filtered_index = false;  // layout 5: the index entries carry a stored size and a filter mask
    uint8_t fill[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // the dataset's fill value as stored, `esize` bytes (what never-written storage reads as)
    size_t fill_size = 0;
    std::vector<uint64_t> chunk; // chunk dims (elements), rank entries
    struct Filter { uint16_t id; };
    std::vector<Filter> filters;
    uint32_t single_filter_mask = 0;
};

Dataset parse_dataset(H5 &h, const std::vector<Message> &msgs)
{
    Dataset d;
    bool have_space = false, have_type = false;
    for (const Message &m : msgs) {
        Cur c(h, m.data.data(), m.data.size());
        if (m.type == 0x0001) { // dataspace
            const uint64_t ver = c.u(1), rank = c.u(1), fl = c.u(1);
            if (ver == 1) c.skip(5);
            else if (ver == 2) c.skip(1);
            else h.fail("dataspace message version %llu", (unsigned long long)ver);
            if (rank > 32) h.fail("dataspace rank %llu", (unsigned long long)rank);
            for (uint64_t i = 0; i < rank; ++i) d.dims.push_back(c.len());
            (void)fl;
            have_space = true;
        } else if (m.type == 0x0003) { // datatype
            const uint64_t cv = c.u(1), b0 = c.u(1);
            c.skip(2);
            const uint64_t size = c.u(4), cls = cv & 0x0f;
            d.big_endian = b0 & 1;
            d.esize = (size_t)size;
            if (cls == 1 && (size == 4 || size == 8)) {
                c.u(2);
                const uint64_t prec = c.u(2), eloc = c.u(1), esz = c.u(1), mloc = c.u(1), msz = c.u(1), bias = c.u(4);
                const bool f32 = size == 4 && prec == 32 && eloc == 23 && esz == 8 && mloc == 0 && msz == 23 && bias == 127;
                const bool f64 = size == 8 && prec == 64 && eloc == 52 && esz == 11 && mloc == 0 && msz == 52 && bias == 1023;
                if (!f32 && !f64) h.fail("variable 'color' is neither IEEE binary32 nor binary64");
                d.is_float = true;
            } else if (cls == 0 && (size == 1 || size == 2 || size == 4 || size == 8)) {
                const uint64_t bit_off = c.u(2), prec = c.u(2);
                if (bit_off != 0 || prec != 8 * size) h.fail("variable 'color' is an integer of %llu bits in %llu bytes", (unsigned long long)prec, (unsigned long long)size);
                d.is_float = false;
                d.is_signed = (b0 & 0x08) != 0;
            } else {
                h.fail("variable 'color' is not a numeric dataset netCDF can hand over as float (datatype class %llu, size %llu)", (unsigned long long)cls, (unsigned long long)size);
            }
            have_type = true;
        } else if (m.type == 0x0008) { // data layout
            const uint64_t ver = c.u(1);
            if (ver == 3) {
                const uint64_t cls = c.u(1);
                if (cls == 1) {
                    d.layout = 1;
                    d.addr = c.off();
                    d.size = c.len();
                } else if (cls == 2) {
                    d.layout = 2;
                    const uint64_t nd = c.u(1);
                    d.addr = c.off();
                    for (uint64_t i = 0; i + 1 < nd; ++i) d.chunk.push_back(c.u(4));
                    d.chunk_elem = c.u(4);
                } else if (cls == 0) {
                    d.layout = 0;
                    d.size = c.u(2);
                    c.need((size_t)d.size);
                    d.compact.assign(c.p, c.p + d.size);
                } else {
                    h.fail("data layout class %llu is not read", (unsigned long long)cls);
                }
            } else if (ver == 4) {
                const uint64_t cls = c.u(1);
                if (cls == 1) {
                    d.layout = 1;
                    d.addr = c.off();
                    d.size = c.len();
                } else if (cls == 0) {
                    d.layout = 0;
                    d.size = c.u(2);
                    c.need((size_t)d.size);
                    d.compact.assign(c.p, c.p + d.size);
                } else if (cls == 2) {
                    const uint64_t fl = c.u(1), nd = c.u(1), enc = c.u(1);
                    if (enc < 1 || enc > 8) h.fail("chunk dimension size encoded in %llu bytes", (unsigned long long)enc);
                    for (uint64_t i = 0; i + 1 < nd; ++i) d.chunk.push_back(c.u((int)enc));
                    d.chunk_elem = c.u((int)enc);
                    const uint64_t idx = c.u(1);
                    if (idx == 1) { // single chunk
                        d.layout = 3;
                        if (fl & 2) { d.size = c.len(); d.single_filter_mask = (uint32_t)c.u(4); }
                    } else if (idx == 2) { // implicit: every chunk allocated, unfiltered, one after the other in row-major chunk order
                        d.layout = 4;
                    } else if (idx == 3) { // fixed array (no unlimited dimension): what HDF5 1.10's "latest" format writes
                        d.layout = 5;
                        c.u(1); // page bits: repeated in the array's header
                    } else {
                        h.fail("version-4 chunk index type %llu is not read (extensible array / version-2 B-tree: datasets with unlimited dimensions)", (unsigned long long)idx);
                    }
                    d.addr = c.off();
                } else {
                    h.fail("data layout class %llu is not read", (unsigned long long)cls);
                }
            } else {
                h.fail("data layout message version %llu is not read (HDF5 older than 1.6.3)", (unsigned long long)ver);
            }
        } else if (m.type == 0x0005) { // fill value: what storage that was never written reads as (netcdf-c defines one for every variable)
            const uint64_t ver = c.u(1);
            bool defined = false;
            if (ver == 1 || ver == 2) {
                c.skip(2);
                defined = c.u(1) != 0;
                if (ver == 1) defined = true; // (version 1 always carries the size field)
            } else if (ver == 3) {
                defined = (c.u(1) & 0x20) != 0;
            }
            if (defined && c.left() >= 4) {
                const uint64_t sz = c.u(4);
                if (sz >= 1 && sz <= 8 && c.left() >= sz) { memcpy(d.fill, c.p, (size_t)sz); d.fill_size = (size_t)sz; }
            }
        } else if (m.type == 0x000B) { // filter pipeline
            const uint64_t ver = c.u(1), n = c.u(1);
            if (n > 32) h.fail("filter pipeline of %llu filters (the format allows 32)", (unsigned long long)n);
            if (ver == 1) c.skip(6);
            else if (ver != 2) h.fail("filter pipeline version %llu", (unsigned long long)ver);
            for (uint64_t i = 0; i < n; ++i) {
                const uint16_t id = (uint16_t)c.u(2);
                uint64_t nlen = 0;
                if (ver == 1 || id >= 256) nlen = c.u(2);
                c.u(2);
                const uint64_t nv = c.u(2);
                if (ver == 1) c.skip((size_t)((nlen + 7) & ~7ull));
                else c.skip((size_t)nlen);
                if (id == 2 && nv >= 1) { c.need(4); d.shuffle_size = (uint32_t)(c.p[0] | c.p[1] << 8 | c.p[2] << 16 | (uint32_t)c.p[3] << 24); }
                c.skip((size_t)nv * 4);
                if (ver == 1 && (nv & 1)) c.skip(4);
                if (id != 1 && id != 2 && id != 3) h.fail("filter %u is not supported (deflate, shuffle, fletcher32 are)", id);
                d.filters.push_back({id});
            }
        }
    }
    if (!have_space || !have_type || d.layout < 0) h.fail("object 'color' is not a dataset (dataspace, datatype or layout message missing)");
    if (d.chunk_elem != 0 && d.chunk_elem != d.esize) h.fail("chunk element size %llu, datatype size %zu", (unsigned long long)d.chunk_elem, d.esize);
    if (d.shuffle_size == 0) d.shuffle_size = (uint32_t)d.esize;
    if (d.shuffle_size > 64) h.fail("shuffle filter over elements of %u bytes", d.shuffle_size);
    if (d.fill_size != 0 && d.fill_size != d.esize) d.fill_size = 0; // (a fill value of another size than the element: not this dataset's)
    return d;
}

// one stored chunk -> `raw` (chunk_bytes) through the inverse filter pipeline
void decode_chunk(H5 &h, const Dataset &d, uint64_t addr, uint64_t stored, uint32_t mask, size_t chunk_bytes, std::vector<uint8_t> &raw,
                  std::vector<uint8_t> &tmp)
{
    if (stored > ((uint64_t)1 << 31) || stored > h.file_size) h.fail("chunk of %llu bytes is not plausible", (unsigned long long)stored);
    tmp.resize((size_t)stored);
    h.read(addr, tmp.data(), (size_t)stored);
    size_t cur = (size_t)stored;
    for (size_t i = d.filters.size(); i-- > 0;) {
        if (mask & (1u << i)) continue;
        const uint16_t id = d.filters[i].id;
        if (id == 3) { // fletcher32: the checksum trails the data
            if (cur < 4) h.fail("chunk too short for its checksum");
            cur -= 4;
        } else if (id == 1) { // deflate
            raw.resize(chunk_bytes);
            uLongf n = (uLongf)chunk_bytes;
            if (uncompress(raw.data(), &n, tmp.data(), (uLong)cur) != Z_OK) h.fail("a chunk does not inflate");
            cur = (size_t)n;
            tmp.assign(raw.begin(), raw.begin() + (long)cur);
        } else if (id == 2) { // shuffle: byte b of every element stored together
            raw.resize(cur);
            const size_t E = d.shuffle_size, n = cur / E;
            if (E > 1) {
                for (size_t e = 0; e < n; ++e)
                    for (size_t b = 0; b < E; ++b) raw[E * e + b] = tmp[b * n + e];
                for (size_t r = E * n; r < cur; ++r) raw[r] = tmp[r];
                tmp.assign(raw.begin(), raw.begin() + (long)cur);
            }
        }
    }
    if (cur != chunk_bytes) h.fail("a chunk decodes to %zu bytes, expected %zu", cur, chunk_bytes);
    raw.assign(tmp.begin(), tmp.begin() + (long)cur);
}

// n stored elements -> f32, as netcdf-c converts on nc_get_vara_float: a C cast of the value (round to nearest for binary64 and for
// integers beyond 2^24)
// (nc4_convert_type counts a binary64 beyond +-FLT_MAX as a range error and the read returns NC_ERANGE, which the reference's `?`
// turns into a failed load: `range_errors` carries that count)
thread_local size_t range_errors = 0;
void to_f32(const Dataset &d, const uint8_t *src, float *dst, size_t n)
{
    const size_t E = d.esize;
    if (d.is_float && E == 4 && !d.big_endian) {
        memcpy(dst, src, n * 4);
        return;
    }
    for (size_t i = 0; i < n; ++i) {
        uint64_t v = 0;
        const uint8_t *p = src + i * E;
        if (d.big_endian) for (size_t b = 0; b < E; ++b) v = (v << 8) | p[b];
        else for (size_t b = 0; b < E; ++b) v |= (uint64_t)p[b] << (8 * b);
        if (d.is_float) {
            if (E == 4) { const uint32_t w = (uint32_t)v; memcpy(dst + i, &w, 4); }
            else {
                double x;
                memcpy(&x, &v, 8);
                if (x > 3.402823466e+38 || x < -3.402823466e+38) ++range_errors; // (infinities included, as netcdf-c's comparison has it; NaN passes)
                dst[i] = (float)x;
            }
        } else if (d.is_signed) {
            const int sh = 64 - 8 * (int)E;
            dst[i] = (float)((int64_t)(v << sh) >> sh);
        } else {
            dst[i] = (float)v;
        }
    }
}

void copy_chunk(const Dataset &d, const std::vector<uint64_t> &origin, const uint8_t *raw, float *out)
{
    // 4-d chunk at `origin` (elements) into the row-major (c, z, y, x) array; edge chunks are stored whole, clipped here
    const uint64_t *D = d.dims.data(), *C = d.chunk.data();
    if (origin[0] >= D[0] || origin[1] >= D[1] || origin[2] >= D[2] || origin[3] >= D[3]) return;
    // (origin < D on every axis and the loops below stop at D: the last element written is inside the (c, z, y, x) array,
    // whose size vrm_hdf5_read_color has checked against the caller's buffer without wrap-around)
    const uint64_t n3 = (C[3] <= D[3] - origin[3]) ? C[3] : D[3] - origin[3];
    for (uint64_t a = 0; a < C[0] && origin[0] + a < D[0]; ++a)
        for (uint64_t b = 0; b < C[1] && origin[1] + b < D[1]; ++b)
            for (uint64_t c = 0; c < C[2] && origin[2] + c < D[2]; ++c) {
                const uint8_t *src = raw + d.esize * (((a * C[1] + b) * C[2] + c) * C[3]);
                float *dst = out + (((origin[0] + a) * D[1] + origin[1] + b) * D[2] + origin[2] + c) * D[3] + origin[3];
                to_f32(d, src, dst, (size_t)n3);
            }
}

void read_chunked(H5 &h, const Dataset &d, float *out)
{
    if (d.chunk.size() != 4) h.fail("chunk rank %zu, expected 4", d.chunk.size());
    size_t chunk_bytes = d.esize;
    for (uint64_t c : d.chunk) {
        if (c == 0 || c > ((uint64_t)1 << 31)) h.fail("bad chunk dimension");
        if (chunk_bytes > ((size_t)1 << 32) / (size_t)c) h.fail("chunk too large");
        chunk_bytes *= (size_t)c;
    }
    // (a chunk larger than the dataset it tiles is not something a writer produces; refusing it keeps a 100-byte file from
    // asking for gigabytes of staging memory)
    for (size_t k = 0; k < 4; ++k)
        if (d.chunk[k] > d.dims[k]) h.fail("chunk dimension %zu is %llu, the dataset's is %llu", k, (unsigned long long)d.chunk[k], (unsigned long long)d.dims[k]);
    std::vector<uint8_t> raw, tmp;
    if (d.layout == 3) {
        if (d.addr == UNDEF) return; // never written: fill value 0
        decode_chunk(h, d, d.addr, d.filters.empty() ? chunk_bytes : d.size, d.single_filter_mask, chunk_bytes, raw, tmp);
        copy_chunk(d, {0, 0, 0, 0}, raw.data(), out);
        return;
    }
    if (d.addr == UNDEF) return;
    if (d.layout == 4 || d.layout == 5) {
        // version-4 layouts index the chunks by their number in row-major order over the chunk grid
        uint64_t grid[4], total = 1;
        for (size_t k = 0; k < 4; ++k) {
            grid[k] = (d.dims[k] + d.chunk[k] - 1) / d.chunk[k];
            if (__builtin_mul_overflow(total, grid[k], &total)) h.fail("chunk grid is not plausible");
        }
        if (total > h.file_size) h.fail("%llu chunks in a file of %llu bytes", (unsigned long long)total, (unsigned long long)h.file_size);
        auto origin_of = [&](uint64_t i) {
            std::vector<uint64_t> o(4);
            for (int k = 3; k >= 0; --k) { o[(size_t)k] = (i % grid[k]) * d.chunk[(size_t)k]; i /= grid[k]; }
            return o;
        };
        if (d.layout == 4) {
            if (!d.filters.empty()) h.fail("implicit chunk index on a filtered dataset");
            for (uint64_t i = 0; i < total; ++i) {
                decode_chunk(h, d, d.addr + i * chunk_bytes, chunk_bytes, 0, chunk_bytes, raw, tmp);
                copy_chunk(d, origin_of(i), raw.data(), out);
            }
            return;
        }
        // fixed array: header "FAHD" -> data block "FADB", its elements in pages once there are more than 2^page_bits of them
        auto hb = h.block(d.addr, (size_t)(8 + h.sl + h.so + 4));
        if (memcmp(hb.data(), "FAHD", 4) != 0) h.fail("bad fixed-array header signature");
        Cur hc(h, hb.data() + 4, hb.size() - 4);
        if (hc.u(1) != 0) h.fail("fixed-array header version");
        const uint64_t client = hc.u(1), esz = hc.u(1), page_bits = hc.u(1), nelem = hc.len(), dblk = hc.off();
        const bool filtered = client == 1;
        if (client > 1 || nelem < total || page_bits > 30) h.fail("fixed array of %llu entries for %llu chunks", (unsigned long long)nelem, (unsigned long long)total);
        if (esz < (uint64_t)h.so + (filtered ? 5 : 0) || esz > (uint64_t)h.so + 12) h.fail("fixed-array entries of %llu bytes", (unsigned long long)esz);
        if (filtered == d.filters.empty()) h.fail("fixed-array client does not match the filter pipeline");
        if (dblk == UNDEF) return; // nothing written yet
        const uint64_t per_page = (uint64_t)1 << page_bits;
        const bool paged = nelem > per_page;
        const uint64_t npages = paged ? (nelem + per_page - 1) / per_page : 0, bitmap_bytes = (npages + 7) / 8;
        const uint64_t prefix = 4 + 1 + 1 + (uint64_t)h.so; // signature, version, client, header address
        auto db = h.block(dblk, (size_t)(prefix + bitmap_bytes));
        if (memcmp(db.data(), "FADB", 4) != 0) h.fail("bad fixed-array data block signature");
        auto element = [&](const uint8_t *e, uint64_t i) {
            Cur ec(h, e, (size_t)esz);
            const uint64_t addr = ec.off();
            uint64_t stored = chunk_bytes;
            uint32_t mask = 0;
            if (filtered) { stored = ec.u((int)(esz - (uint64_t)h.so - 4)); mask = (uint32_t)ec.u(4); }
            if (addr == UNDEF) return; // never written: the fill value
            decode_chunk(h, d, addr, stored, mask, chunk_bytes, raw, tmp);
            copy_chunk(d, origin_of(i), raw.data(), out);
        };
        if (!paged) {
            auto eb = h.block(dblk + prefix, (size_t)(total * esz));
            for (uint64_t i = 0; i < total; ++i) element(eb.data() + i * esz, i);
        } else {
            // (pages follow the data block's prefix, bitmap and checksum; each is 2^page_bits entries -- the last one what is left --
            // and a checksum; a page whose bit is clear was never written)
            uint64_t at = dblk + prefix + bitmap_bytes + 4;
            for (uint64_t pg = 0; pg < npages; ++pg) {
                const uint64_t first = pg * per_page, cnt = std::min<uint64_t>(per_page, nelem - first);
                if (db[(size_t)(prefix + pg / 8)] & (0x80u >> (pg % 8))) {   // (most significant bit first)
                    auto eb = h.block(at, (size_t)(cnt * esz));
                    for (uint64_t i = 0; i < cnt && first + i < total; ++i) element(eb.data() + i * esz, first + i);
                    at += cnt * esz + 4;
                }
            }
        }
        return;
    }
    std::vector<uint64_t> nodes{d.addr};
    size_t visited = 0;
    const size_t key = 8 + 8 * 5; // chunk size, filter mask, 5 offsets of 8 bytes (rank + 1)
    while (!nodes.empty()) {
        if (++visited > 50000000) h.fail("chunk B-tree is not plausible");
        const uint64_t a = nodes.back();
        nodes.pop_back();
        auto nb = h.block(a, (size_t)(8 + 2 * h.so));
        if (memcmp(nb.data(), "TREE", 4) != 0) h.fail("bad chunk B-tree node signature");
        Cur nc(h, nb.data() + 4, nb.size() - 4);
        if (nc.u(1) != 1) h.fail("chunk B-tree node has the wrong type");
        const uint64_t level = nc.u(1), used = nc.u(2);
        auto body = h.block(a + 8 + 2 * (uint64_t)h.so, (size_t)(used * (key + h.so) + key));
        Cur bc(h, body.data(), body.size());
        for (uint64_t i = 0; i < used; ++i) {
            const uint64_t stored = bc.u(4);
            const uint32_t mask = (uint32_t)bc.u(4);
            std::vector<uint64_t> origin(4);
            for (int k = 0; k < 4; ++k) origin[(size_t)k] = bc.u(8);
            bc.u(8);
            const uint64_t child = bc.off();
            if (level > 0) {
                nodes.push_back(child);
            } else {
                decode_chunk(h, d, child, stored, mask, chunk_bytes, raw, tmp);
                copy_chunk(d, origin, raw.data(), out);
            }
        }
    }
}

} // namespace

// `f` is open on a file that starts with the HDF5 signature.  With out == NULL a size query.
int vrm_hdf5_read_color(FILE *f, const char *path, uint32_t *x, uint32_t *y, uint32_t *z, float *out, size_t capacity_floats)
{
    try {
        H5 h{f, path};
        if (fseeko(f, 0, SEEK_END) != 0) h.fail("cannot seek");
        h.file_size = (uint64_t)ftello(f);
        // the superblock sits at 0, 512, 1024, ... (netCDF-4 files: 0)
        uint64_t sb = UNDEF;
        static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
        for (uint64_t a = 0; a + 8 <= h.file_size && a <= ((uint64_t)1 << 30); a = a ? 2 * a : 512) {
            uint8_t s[8];
            h.read(a, s, 8);
            if (memcmp(s, sig, 8) == 0) { sb = a; break; }
        }
        if (sb == UNDEF) h.fail("neither a NetCDF classic file (CDF\\001/\\002/\\005) nor a NetCDF-4 / HDF5 file (no superblock signature)");
        auto b = h.block(sb, (size_t)std::min<uint64_t>(h.file_size - sb, 128));
        Cur c(h, b.data(), b.size());
        c.skip(8);
        const uint64_t ver = c.u(1);
        uint64_t root = UNDEF;
        std::vector<Message> root_msgs;
        if (ver <= 1) {
            c.skip(4); // free-space version, root group version, reserved, shared header version
            h.so = (int)c.u(1);
            h.sl = (int)c.u(1);
            c.skip(1);
            c.skip(4); // group leaf / internal node K
            c.skip(4); // consistency flags
            if (ver == 1) c.skip(4);
            if ((h.so != 4 && h.so != 8) || (h.sl != 4 && h.sl != 8)) h.fail("offsets of %d and lengths of %d bytes are not read", h.so, h.sl);
            const uint64_t base = c.off();
            c.off();
            c.off();
            c.off();
            h.base = base == UNDEF ? 0 : base;
            c.off(); // root symbol table entry: link name offset
            root = c.off();
        } else if (ver <= 3) {
            h.so = (int)c.u(1);
            h.sl = (int)c.u(1);
            c.skip(1);
            if ((h.so != 4 && h.so != 8) || (h.sl != 4 && h.sl != 8)) h.fail("offsets of %d and lengths of %d bytes are not read", h.so, h.sl);
            const uint64_t base = c.off();
            c.off();
            c.off();
            root = c.off();
            h.base = base == UNDEF ? 0 : base;
        } else {
            h.fail("superblock version %llu is not read", (unsigned long long)ver);
        }
        root_msgs = object_header(h, root);
        Links links;
        group_links(h, root_msgs, links);
        uint64_t color = UNDEF;
        uint64_t named[3] = {UNDEF, UNDEF, UNDEF};
        for (auto &l : links) {
            if (l.first == "color") color = l.second;
            if (l.first == "x") named[0] = l.second;
            if (l.first == "y") named[1] = l.second;
            if (l.first == "z") named[2] = l.second;
        }
        if (color == UNDEF) h.fail("could not find 'color' variable"); // netcdf.rs:37 expect()
        const Dataset d = parse_dataset(h, object_header(h, color));
        if (d.dims.size() != 4) h.fail("variable 'color' has %zu dimensions, expected (c, z, y, x)", d.dims.size());
        if (d.dims[0] != 4) h.fail("dimension c is %llu, expected 4", (unsigned long long)d.dims[0]);
        for (int i = 1; i < 4; ++i)
            if (d.dims[(size_t)i] == 0 || d.dims[(size_t)i] > 0xffffffffull) h.fail("dimension %d of 'color' is %llu", i, (unsigned long long)d.dims[(size_t)i]);
        // netcdf.rs:39-41 takes the lengths from the dimensions named x, y, z: where those dimension scales exist they must agree
        for (int i = 0; i < 3; ++i)
            if (named[i] != UNDEF) {
                try {
                    const Dataset s = parse_dataset(h, object_header(h, named[i]));
                    if (s.dims.size() == 1 && s.dims[0] != d.dims[(size_t)(3 - i)])
                        h.fail("dimension '%c' has length %llu but 'color' has %llu there: not color(c,z,y,x)", "xyz"[i],
                               (unsigned long long)s.dims[0], (unsigned long long)d.dims[(size_t)(3 - i)]);
                } catch (const Fail &e) {
                    if (e.msg.find("not color(c,z,y,x)") != std::string::npos) throw;
                    // a dimension scale of a type this reader does not parse: its length is not needed
                }
            }
        *z = (uint32_t)d.dims[1];
        *y = (uint32_t)d.dims[2];
        *x = (uint32_t)d.dims[3];
        {
            uint64_t t = 0; // (a caller sizes its buffer as 4*x*y*z from this query: refuse dimensions whose product wraps)
            if (__builtin_mul_overflow(d.dims[1], d.dims[2], &t) || __builtin_mul_overflow(t, d.dims[3], &t) || t > (uint64_t)SIZE_MAX / 16)
                h.fail("color(4,%llu,%llu,%llu) does not fit the address space", (unsigned long long)d.dims[1], (unsigned long long)d.dims[2], (unsigned long long)d.dims[3]);
        }
        if (!out) return VRM_OK;
        // 4 * z * y * x floats: each dimension is < 2^32, so the product of three can wrap 64 bits -- checked, not assumed
        uint64_t n = 0;
        if (__builtin_mul_overflow(d.dims[1], d.dims[2], &n) || __builtin_mul_overflow(n, d.dims[3], &n) || __builtin_mul_overflow(n, (uint64_t)4, &n) ||
            n > (uint64_t)SIZE_MAX / 4)
            h.fail("color(4,%llu,%llu,%llu) does not fit the address space", (unsigned long long)d.dims[1], (unsigned long long)d.dims[2], (unsigned long long)d.dims[3]);
        if (capacity_floats < n) {
            vrm_set_error("%s holds %llu floats, buffer holds %zu", path, (unsigned long long)n, capacity_floats);
            return VRM_ERR_INVALID;
        }
        const uint64_t E = d.esize;
        range_errors = 0;
        auto fill_all = [&] { // storage that was never written reads as the dataset's fill value (zero bytes unless the file defines one)
            float v = 0.0f;
            if (d.fill_size) to_f32(d, d.fill, &v, 1);
            uint32_t w;
            memcpy(&w, &v, 4);
            if (w == 0) memset(out, 0, (size_t)n * 4);
            else for (uint64_t i = 0; i < n; ++i) memcpy(out + i, &w, 4);
        };
        if (d.layout == 0) {
            if (d.compact.size() < n * E) h.fail("compact storage of %zu bytes is shorter than the %llu the dimensions need", d.compact.size(), (unsigned long long)(n * E));
            to_f32(d, d.compact.data(), out, (size_t)n);
        } else if (d.layout == 1) {
            if (d.addr == UNDEF) {
                fill_all();
            } else {
                if (d.size < n * E) h.fail("contiguous storage of %llu bytes is shorter than the %llu the dimensions need", (unsigned long long)d.size, (unsigned long long)(n * E));
                const uint64_t step = (uint64_t)1 << 24; // elements per block
                if (d.is_float && E == 4 && !d.big_endian) {
                    for (uint64_t done = 0; done < n; done += step) h.read(d.addr + done * 4, reinterpret_cast<uint8_t *>(out + done), (size_t)(std::min<uint64_t>(step, n - done) * 4));
                } else {
                    std::vector<uint8_t> blk((size_t)(std::min<uint64_t>(step, n) * E));
                    for (uint64_t done = 0; done < n; done += step) {
                        const uint64_t k = std::min<uint64_t>(step, n - done);
                        h.read(d.addr + done * E, blk.data(), (size_t)(k * E));
                        to_f32(d, blk.data(), out + done, (size_t)k);
                    }
                }
            }
        } else {
            fill_all(); // chunks that were never written
            read_chunked(h, d, out);
        }
        if (range_errors) h.fail("%zu values of 'color' are binary64 beyond the range of f32: netcdf-c reads them as NC_ERANGE and the reference's load fails", range_errors);
        return VRM_OK;
    } catch (const Fail &e) {
        vrm_set_error("%s", e.msg.c_str());
        return VRM_ERR_IO;
    }
}

// ---- writer --------------------------------------------------------------------------------------------------------
// loaders::netcdf::write_voxel_grid (loaders/netcdf.rs:8-31) creates its file with `netcdf::create`, i.e. a NetCDF-4 /
// HDF5 container.  This writes that container in the oldest dialect every HDF5 library reads (the one netcdf-c itself
// produced before 4.x turned creation-order tracking on): superblock version 0, version-1 object headers, the root group
// as a symbol table (one B-tree node, one symbol node, a local heap), `color(c,z,y,x)` a contiguous little-endian f32
// dataset, and the four dimensions c, z, y, x as dimension-scale datasets without storage that carry netcdf-c's
// attributes (CLASS, NAME, _Netcdf4Dimid); `color` carries _Netcdf4Coordinates = {0,1,2,3}, the attribute netcdf-c
// reads the dimension ids of a variable from, so no object references (DIMENSION_LIST / REFERENCE_LIST) are written.
// PINNED against the HDF5 library (tests/test_hdf5_golden_cpu.py): libhdf5 1.10.6's own h5dump opens the result, lists the five
// datasets with their shapes, types and attributes and returns the values that went in -- which is how the one place where
// the library departs from the published format was found (the local heap's empty free list is stored as 1, not as the
// undefined address).  Also read back bit for bit by the reader above, and assembled byte for byte by tests/hdf5_builder.py,
// a second restatement in another language.  netcdf-c itself is not available here: that it takes the dimension ids from
// `_Netcdf4Coordinates` is from its source, not from a run.
namespace {

struct Bytes {
    std::vector<uint8_t> b;
    void u(uint64_t v, int n) { for (int i = 0; i < n; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
    void raw(const void *p, size_t n) { const uint8_t *q = static_cast<const uint8_t *>(p); b.insert(b.end(), q, q + n); }
    void zeros(size_t n) { b.insert(b.end(), n, 0); }
    void pad8() { zeros((8 - b.size() % 8) % 8); }
    void cat(const Bytes &o) { b.insert(b.end(), o.b.begin(), o.b.end()); }
};

typedef std::pair<int, Bytes> Msg; // (message type, body)

Bytes dataspace_v1(const std::vector<uint64_t> &dims)
{
    Bytes m;
    m.u(1, 1); m.u(dims.size(), 1); m.u(0, 1); m.u(0, 1); m.zeros(4);
    for (uint64_t d : dims) m.u(d, 8);
    return m;
}
Bytes datatype_f32(bool big_endian)
{
    Bytes m; // class 1 version 1; byte order, mantissa normalisation 2 (implied msb); sign bit 31; 4 bytes
    m.u(0x11, 1); m.u((big_endian ? 1 : 0) | (2 << 4), 1); m.u(31, 1); m.u(0, 1); m.u(4, 4);
    m.u(0, 2); m.u(32, 2); m.u(23, 1); m.u(8, 1); m.u(0, 1); m.u(23, 1); m.u(127, 4); // bit offset, precision, exponent at 23 (8 bits), mantissa at 0 (23 bits), bias
    return m;
}
Bytes datatype_i32()
{
    Bytes m; // class 0 version 1; little-endian, signed
    m.u(0x10, 1); m.u(0x08, 1); m.u(0, 1); m.u(0, 1); m.u(4, 4); m.u(0, 2); m.u(32, 2);
    return m;
}
Bytes datatype_string(size_t n)
{
    Bytes m; // class 3 version 1; null-terminated ASCII of n bytes
    m.u(0x13, 1); m.u(0, 1); m.u(0, 1); m.u(0, 1); m.u(n, 4);
    return m;
}
Bytes layout_contiguous(uint64_t addr, uint64_t size)
{
    Bytes m;
    m.u(3, 1); m.u(1, 1); m.u(addr, 8); m.u(size, 8);
    return m;
}
Bytes fill_value_undefined()
{
    Bytes m; // version 2, space allocated late, fill written if set, no fill value defined
    m.u(2, 1); m.u(2, 1); m.u(2, 1); m.u(0, 1);
    return m;
}
Bytes attribute_v1(const char *name, Bytes type, Bytes space, const Bytes &value)
{
    const size_t nn = strlen(name) + 1;
    Bytes m;
    m.u(1, 1); m.u(0, 1); m.u(nn, 2); m.u(type.b.size(), 2); m.u(space.b.size(), 2);
    m.raw(name, nn); m.pad8();
    type.pad8(); m.cat(type);
    space.pad8(); m.cat(space);
    m.cat(value);
    return m;
}
Bytes attribute_string(const char *name, const std::string &text)
{
    Bytes v;
    v.raw(text.c_str(), text.size() + 1); // the terminator belongs to the value
    return attribute_v1(name, datatype_string(text.size() + 1), dataspace_v1({}), v);
}
Bytes attribute_i32(const char *name, const std::vector<int32_t> &values, bool scalar)
{
    Bytes v;
    for (int32_t x : values) v.u((uint32_t)x, 4);
    return attribute_v1(name, datatype_i32(), scalar ? dataspace_v1({}) : dataspace_v1({values.size()}), v);
}

// appends a version-1 object header, returns its offset in `img`
size_t object_header_v1(Bytes &img, const std::vector<Msg> &msgs)
{
    Bytes body;
    for (const Msg &m : msgs) {
        Bytes d = m.second;
        d.pad8();
        body.u((uint64_t)m.first, 2); body.u(d.b.size(), 2); body.u(0, 1); body.zeros(3);
        body.cat(d);
    }
    img.pad8();
    const size_t at = img.b.size();
    img.u(1, 1); img.u(0, 1); img.u(msgs.size(), 2); img.u(1, 4); img.u(body.b.size(), 4); img.zeros(4);
    img.cat(body);
    return at;
}

const int GROUP_LEAF_K = 4, GROUP_INTERNAL_K = 16;

} // namespace

int vrm_hdf5_write_color(const char *path, uint32_t x, uint32_t y, uint32_t z, const float *color_czyx)
{
    static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
    const uint64_t n = (uint64_t)4 * x * y * z, bytes = n * 4;
    const uint64_t data_addr = 96;         // right behind the superblock
    const uint64_t base = data_addr + bytes; // metadata behind the data (a multiple of 8)
    Bytes img;
    const char *all[5] = {"c", "z", "y", "x", "color"}; // netcdf.rs:11-16
    const uint64_t lens[4] = {4, z, y, x};
    uint64_t obj[5];
    for (int i = 0; i < 4; ++i) {
        char text[80];
        snprintf(text, sizeof text, "This is a netCDF dimension but not a netCDF variable.%10llu", (unsigned long long)lens[i]);
        obj[i] = base + object_header_v1(img, {{0x01, dataspace_v1({lens[i]})}, {0x03, datatype_f32(true)}, {0x05, fill_value_undefined()},
                                               {0x08, layout_contiguous(UNDEF, lens[i] * 4)}, {0x0C, attribute_string("CLASS", "DIMENSION_SCALE")},
                                               {0x0C, attribute_string("NAME", text)}, {0x0C, attribute_i32("_Netcdf4Dimid", {i}, true)}});
    }
    obj[4] = base + object_header_v1(img, {{0x01, dataspace_v1({4, z, y, x})}, {0x03, datatype_f32(false)}, {0x05, fill_value_undefined()}, // netcdf.rs:16
                                           {0x08, layout_contiguous(data_addr, bytes)}, {0x0C, attribute_i32("_Netcdf4Coordinates", {0, 1, 2, 3}, false)}});
    // the root group's symbol table: entries sorted by name
    const int order[5] = {0, 4, 3, 2, 1}; // "c" < "color" < "x" < "y" < "z"
    Bytes heap; // offset 0: the empty name of the B-tree's first key
    heap.zeros(8);
    uint64_t name_off[5];
    for (int k = 0; k < 5; ++k) {
        name_off[order[k]] = heap.b.size();
        heap.raw(all[order[k]], strlen(all[order[k]]) + 1);
        heap.pad8();
    }
    img.pad8();
    const uint64_t heap_data = base + img.b.size();
    img.cat(heap);
    img.pad8();
    const uint64_t heap_addr = base + img.b.size();
    img.raw("HEAP", 4); img.u(0, 1); img.zeros(3); img.u(heap.b.size(), 8); img.u(1, 8); img.u(heap_data, 8); // (free list empty: the library writes 1 here, H5HL_FREE_NULL, not the undefined address)
    img.pad8();
    const uint64_t snod = base + img.b.size();
    img.raw("SNOD", 4); img.u(1, 1); img.u(0, 1); img.u(5, 2);
    for (int k = 0; k < 5; ++k) { img.u(name_off[order[k]], 8); img.u(obj[order[k]], 8); img.u(0, 4); img.u(0, 4); img.zeros(16); }
    img.zeros((size_t)(2 * GROUP_LEAF_K - 5) * 40); // a symbol node always has room for 2K entries
    const uint64_t tree = base + img.b.size();
    img.raw("TREE", 4); img.u(0, 1); img.u(0, 1); img.u(1, 2); img.u(UNDEF, 8); img.u(UNDEF, 8);
    img.u(0, 8); img.u(snod, 8); img.u(name_off[order[4]], 8);
    img.zeros((size_t)(2 * GROUP_INTERNAL_K - 1) * 16); // ... and a B-tree node for 2K children and 2K+1 keys
    const uint64_t root = base + object_header_v1(img, {{0x11, [&] { Bytes m; m.u(tree, 8); m.u(heap_addr, 8); return m; }()},
                                                        {0x0C, attribute_string("_NCProperties", "version=2,volume_renderer_amd=3")}});
    const uint64_t eof = base + img.b.size();
    Bytes sb;
    sb.raw(sig, 8);
    sb.u(0, 1); sb.u(0, 1); sb.u(0, 1); sb.u(0, 1); sb.u(0, 1); sb.u(8, 1); sb.u(8, 1); sb.u(0, 1); // versions; offsets and lengths of 8 bytes
    sb.u(GROUP_LEAF_K, 2); sb.u(GROUP_INTERNAL_K, 2); sb.u(0, 4);
    sb.u(0, 8); sb.u(UNDEF, 8); sb.u(eof, 8); sb.u(UNDEF, 8); // base, free-space info, end of file, driver info
    sb.u(0, 8); sb.u(root, 8); sb.u(1, 4); sb.u(0, 4); sb.u(tree, 8); sb.u(heap_addr, 8); // root entry, its scratch = B-tree and heap
    if (sb.b.size() != data_addr) {
        vrm_set_error("internal: superblock of %zu bytes", sb.b.size());
        return VRM_ERR_STATE;
    }
    FILE *f = fopen(path, "wb");
    if (!f) {
        vrm_set_error("cannot create %s", path);
        return VRM_ERR_IO;
    }
    bool ok = fwrite(sb.b.data(), 1, sb.b.size(), f) == sb.b.size();
    const size_t step = (size_t)1 << 24;
    for (uint64_t done = 0; ok && done < n; done += step) { // host floats are little-endian: stored as they are
        const size_t k = (size_t)std::min<uint64_t>(step, n - done);
        ok = fwrite(color_czyx + done, 4, k, f) == k;
    }
    ok = ok && fwrite(img.b.data(), 1, img.b.size(), f) == img.b.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        vrm_set_error("%s: write failed", path);
        return VRM_ERR_IO;
    }
    return VRM_OK;
}
