// vrm_netcdf.cpp -- the NetCDF interchange of the path: loaders/netcdf.rs:8-69 and
// scripts/diffdvr.py:276-287 exchange volumes as one variable `color(c,z,y,x)` of f32 with c = 4.
//
// This file: the netCDF CLASSIC container (CDF-1, CDF-2 "64-bit offset", CDF-5 "64-bit data"), restated from the
// published file-format specification (NetCDF Users Guide, "File Format Specification": big-endian header of
// dimension / attribute / variable lists, fixed-size variable data at `begin`), reader and writer.  The reference's own
// files are NetCDF-4, i.e. HDF5 containers (netcdf::create in the Rust crate, format='NETCDF4_CLASSIC' in diffdvr.py):
// those are recognised by their signature and read by vrm_hdf5.cpp.  What this library WRITES is the classic container,
// which netcdf-c -- and so the reference's netcdf::open -- reads transparently.  The data model (names, dimension order,
// f32, value range) is the reference's either way.  Host code only; the device side is vrm_grid_upload_f32 /
// vrm_grid_download_rgba8.
#include <sys/stat.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "vrm_internal.h"

// vrm_hdf5.cpp: `color(c,z,y,x)` from a NetCDF-4 / HDF5 container
int vrm_hdf5_read_color(FILE *f, const char *path, uint32_t *x, uint32_t *y, uint32_t *z, float *out, size_t capacity_floats);
int vrm_hdf5_write_color(const char *path, uint32_t x, uint32_t y, uint32_t z, const float *color_czyx);

#define REQUIRE_IO(cond, ...)                             \
    do {                                                  \
        if (!(cond)) {                                    \
            vrm_set_error(__VA_ARGS__);                   \
            return VRM_ERR_IO;                            \
        }                                                 \
    } while (0)

namespace {

enum { NC_DIMENSION = 0x0A, NC_VARIABLE = 0x0B, NC_ATTRIBUTE = 0x0C, NC_FLOAT = 5 };

struct File {
    FILE *f = nullptr;
    ~File() { if (f) fclose(f); }
};

struct Reader {
    FILE *f;
    int version;      // 1, 2 or 5
    bool ok = true;
    uint32_t u32()
    {
        uint8_t b[4];
        if (fread(b, 1, 4, f) != 4) { ok = false; return 0; }
        return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
    }
    uint64_t u64() { const uint64_t hi = u32(); return (hi << 32) | u32(); }
    uint64_t count() { return version == 5 ? u64() : u32(); }   // NON_NEG: INT in CDF-1/2, INT64 in CDF-5
    uint64_t offset() { return version == 1 ? u32() : u64(); }  // OFFSET: INT in CDF-1, INT64 otherwise
    std::string name()
    {
        const uint64_t n = count();
        if (!ok || n > 4096) { ok = false; return std::string(); }
        std::string s((size_t)n, '\0');
        if (n && fread(&s[0], 1, (size_t)n, f) != n) ok = false;
        skip((4 - n % 4) % 4);
        return s;
    }
    void skip(uint64_t n) { if (n && fseeko(f, (off_t)n, SEEK_CUR) != 0) ok = false; }
    static uint64_t type_size(uint32_t t)
    {
        switch (t) {
        case 1: case 2: case 7: return 1;   // byte, char, ubyte
        case 3: case 8: return 2;           // short, ushort
        case 4: case 5: case 9: return 4;   // int, float, uint
        case 6: case 10: case 11: return 8; // double, int64, uint64
        default: return 0;
        }
    }
    void att_list()
    {
        const uint32_t tag = u32();
        const uint64_t n = count();
        if (!ok) return;
        if (tag == 0 && n == 0) return; // ABSENT
        if (tag != NC_ATTRIBUTE) { ok = false; return; }
        for (uint64_t i = 0; i < n && ok; ++i) {
            name();
            const uint32_t t = u32();
            const uint64_t ne = count(), sz = type_size(t);
            if (sz == 0) { ok = false; return; }
            const uint64_t bytes = ne * sz;
            skip(bytes + (4 - bytes % 4) % 4);
        }
    }
};

struct Header {
    uint64_t x = 0, y = 0, z = 0;
    uint64_t begin = 0;
    uint32_t type = NC_FLOAT; // nc_type of `color`
};

int read_header(FILE *f, const char *path, Header *h)
{
    uint8_t magic[8] = {0};
    REQUIRE_IO(fread(magic, 1, 4, f) == 4, "%s: too short for a NetCDF file", path);
    REQUIRE_IO(magic[0] == 'C' && magic[1] == 'D' && magic[2] == 'F' && (magic[3] == 1 || magic[3] == 2 || magic[3] == 5),
               "%s is not a NetCDF classic file (no CDF\\001, CDF\\002 or CDF\\005 signature)", path);
    Reader r{f, magic[3]};
    r.count(); // numrecs
    std::vector<std::string> dim_names;
    std::vector<uint64_t> dim_len;
    {
        const uint32_t tag = r.u32();
        const uint64_t n = r.count();
        REQUIRE_IO(r.ok && ((tag == 0 && n == 0) || tag == NC_DIMENSION) && n <= 1024, "%s: bad dimension list", path);
        for (uint64_t i = 0; i < n && r.ok; ++i) {
            dim_names.push_back(r.name());
            dim_len.push_back(r.count());
        }
    }
    r.att_list(); // global attributes
    REQUIRE_IO(r.ok, "%s: truncated or malformed header", path);
    const uint32_t tag = r.u32();
    const uint64_t nvars = r.count();
    REQUIRE_IO(r.ok && ((tag == 0 && nvars == 0) || tag == NC_VARIABLE) && nvars <= 65536, "%s: bad variable list", path);
    bool found = false;
    for (uint64_t v = 0; v < nvars && r.ok; ++v) {
        const std::string name = r.name();
        const uint64_t nd = r.count();
        REQUIRE_IO(r.ok && nd <= 1024, "%s: bad variable header", path);
        std::vector<uint64_t> ids;
        for (uint64_t i = 0; i < nd; ++i) ids.push_back(r.version == 5 ? r.u64() : r.u32());
        r.att_list();
        const uint32_t type = r.u32();
        r.count();                       // vsize (unreliable beyond 4 GiB by specification: not used)
        const uint64_t begin = r.offset();
        if (name == "color" && r.ok) {   // netcdf.rs:37
            // (netcdf.rs:45 reads with get_values::<f32>: netcdf-c converts every numeric external type; text it refuses, NC_ECHAR)
            REQUIRE_IO(type != 2 && Reader::type_size(type) != 0, "%s: variable 'color' is not numeric (nc_type %u)", path, type);
            h->type = type;
            REQUIRE_IO(nd == 4, "%s: variable 'color' has %llu dimensions, expected (c, z, y, x)", path, (unsigned long long)nd);
            const char *want[4] = {"c", "z", "y", "x"};
            for (int i = 0; i < 4; ++i) {
                REQUIRE_IO(ids[i] < dim_names.size() && dim_names[ids[i]] == want[i],
                           "%s: dimension %d of 'color' is not '%s' (netcdf.rs:16: color(c,z,y,x))", path, i, want[i]);
                REQUIRE_IO(dim_len[ids[i]] > 0, "%s: 'color' uses the unlimited dimension; record variables are not supported", path);
            }
            REQUIRE_IO(dim_len[ids[0]] == 4, "%s: dimension c is %llu, expected 4", path, (unsigned long long)dim_len[ids[0]]);
            h->z = dim_len[ids[1]];
            h->y = dim_len[ids[2]];
            h->x = dim_len[ids[3]];
            h->begin = begin;
            found = true;
        }
    }
    REQUIRE_IO(r.ok, "%s: truncated or malformed header", path);
    REQUIRE_IO(found, "%s: could not find 'color' variable", path); // netcdf.rs:37 expect()
    REQUIRE_IO(h->x <= 0xffffffffull && h->y <= 0xffffffffull && h->z <= 0xffffffffull, "%s: dimensions too large", path);
    return VRM_OK;
}

inline uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

struct Writer {
    std::vector<uint8_t> b;
    int version;
    void u32(uint32_t v) { for (int s = 24; s >= 0; s -= 8) b.push_back((uint8_t)(v >> s)); }
    void u64(uint64_t v) { u32((uint32_t)(v >> 32)); u32((uint32_t)v); }
    void count(uint64_t v) { if (version == 5) u64(v); else u32((uint32_t)v); }
    void offset(uint64_t v) { if (version == 1) u32((uint32_t)v); else u64(v); }
    void name(const char *s)
    {
        const size_t n = strlen(s);
        count(n);
        b.insert(b.end(), s, s + n);
        for (size_t i = n; i % 4; ++i) b.push_back(0);
    }
};

} // namespace

static bool mul3_overflows(uint64_t a, uint64_t b, uint64_t c)
{
    uint64_t t = 0;
    return __builtin_mul_overflow(a, b, &t) || __builtin_mul_overflow(t, c, &t) || t > (uint64_t)SIZE_MAX / 16;
}

extern "C" int vrm_netcdf_read_color(const char *path, uint32_t *x, uint32_t *y, uint32_t *z, float *color_czyx, size_t capacity_floats)
{
    return vrm_guard("vrm_netcdf_read_color", VRM_ERR_IO, [&]() -> int {
    if (!path || !x || !y || !z) {
        vrm_set_error("null argument");
        return VRM_ERR_INVALID;
    }
    File fl;
    fl.f = fopen(path, "rb");
    REQUIRE_IO(fl.f != nullptr, "cannot open %s", path);
    uint8_t sig[4] = {0, 0, 0, 0};
    const size_t got = fread(sig, 1, 4, fl.f);
    // a NetCDF-4 file is an HDF5 container (its superblock may also sit at 512, 1024, ...: anything that is not CDF goes there)
    if (got == 4 && memcmp(sig, "CDF", 3) != 0) return vrm_hdf5_read_color(fl.f, path, x, y, z, color_czyx, capacity_floats);
    rewind(fl.f);
    Header h;
    const int rc = read_header(fl.f, path, &h);
    if (rc) return rc;
    *x = (uint32_t)h.x;
    *y = (uint32_t)h.y;
    *z = (uint32_t)h.z;
    struct stat st;
    REQUIRE_IO(fstat(fileno(fl.f), &st) == 0 && (uint64_t)st.st_size >= h.begin &&
                   !mul3_overflows(h.x, h.y, h.z) && ((uint64_t)st.st_size - h.begin) / (4 * Reader::type_size(h.type)) >= h.x * h.y * h.z,   // (x, y, z < 2^32 each: the product of THREE can wrap 64 bits)
               "%s: data of 'color' is truncated: the header promises color(4,%llu,%llu,%llu) but the file has only %lld bytes", path, (unsigned long long)h.z,
               (unsigned long long)h.y, (unsigned long long)h.x, (long long)st.st_size);
    if (!color_czyx) return VRM_OK; // size query
    const size_t n = (size_t)4 * h.x * h.y * h.z;
    if (capacity_floats < n) {
        vrm_set_error("%s holds %zu floats, buffer holds %zu", path, n, capacity_floats);
        return VRM_ERR_INVALID;
    }
    REQUIRE_IO(fseeko(fl.f, (off_t)h.begin, SEEK_SET) == 0, "%s: cannot seek to the data of 'color'", path);
    const size_t chunk = (size_t)1 << 22;
    if (h.type == NC_FLOAT) {
        for (size_t done = 0; done < n; done += chunk) {
            const size_t k = n - done < chunk ? n - done : chunk;
            REQUIRE_IO(fread(color_czyx + done, 4, k, fl.f) == k, "%s: data of 'color' is truncated", path);
            uint32_t *w = reinterpret_cast<uint32_t *>(color_czyx + done);
            for (size_t i = 0; i < k; ++i) w[i] = bswap32(w[i]); // XDR: big-endian IEEE
        }
        return VRM_OK;
    }
    // any other numeric type: big-endian on disk, handed over as (float)value like nc_get_vara_float
    const size_t E = (size_t)Reader::type_size(h.type);
    const bool is_double = h.type == 6, is_signed = h.type == 1 || h.type == 3 || h.type == 4 || h.type == 10;
    std::vector<uint8_t> blk(chunk * E);
    size_t range_errors = 0; // (ncx_getn_double_float: a double beyond +-FLT_MAX is NC_ERANGE, and the reference's `?` fails the load)
    for (size_t done = 0; done < n; done += chunk) {
        const size_t k = n - done < chunk ? n - done : chunk;
        REQUIRE_IO(fread(blk.data(), E, k, fl.f) == k, "%s: data of 'color' is truncated", path);
        for (size_t i = 0; i < k; ++i) {
            uint64_t v = 0;
            for (size_t b = 0; b < E; ++b) v = (v << 8) | blk[i * E + b];
            float out;
            if (is_double) {
                double x;
                memcpy(&x, &v, 8);
                if (x > 3.402823466e+38 || x < -3.402823466e+38) ++range_errors;
                out = (float)x;
            }
            else if (is_signed) { const int sh = 64 - 8 * (int)E; out = (float)((int64_t)(v << sh) >> sh); }
            else out = (float)v;
            color_czyx[done + i] = out;
        }
    }
    REQUIRE_IO(range_errors == 0, "%s: %zu values of 'color' are doubles beyond the range of f32 (NC_ERANGE: the reference's load fails)", path, range_errors);
    return VRM_OK;
    });
}

extern "C" int vrm_netcdf_write_color(const char *path, uint32_t x, uint32_t y, uint32_t z, const float *color_czyx)
{
    return vrm_guard("vrm_netcdf_write_color", VRM_ERR_IO, [&]() -> int {
    if (!path || !color_czyx || x == 0 || y == 0 || z == 0) {
        vrm_set_error("null argument or empty volume");
        return VRM_ERR_INVALID;
    }
    const uint64_t n = (uint64_t)4 * x * y * z, bytes = n * 4;
    Writer w;
    // the smallest container that can hold the variable: CDF-1 below 2 GiB, CDF-2 below 4 GiB, CDF-5 beyond
    w.version = bytes + 1024 < ((uint64_t)1 << 31) ? 1 : (bytes < ((uint64_t)1 << 32) - 4 ? 2 : 5);
    w.b = {'C', 'D', 'F', (uint8_t)w.version};
    w.count(0); // numrecs
    w.u32(NC_DIMENSION);
    w.count(4);
    const char *dn[4] = {"c", "z", "y", "x"};
    const uint64_t dl[4] = {4, z, y, x};
    for (int i = 0; i < 4; ++i) { // netcdf.rs:11-14
        w.name(dn[i]);
        w.count(dl[i]);
    }
    w.u32(0); w.count(0); // no global attributes
    w.u32(NC_VARIABLE);
    w.count(1);
    w.name("color"); // netcdf.rs:16
    w.count(4);
    for (uint64_t i = 0; i < 4; ++i) { if (w.version == 5) w.u64(i); else w.u32((uint32_t)i); }
    w.u32(0); w.count(0); // no variable attributes
    w.u32(NC_FLOAT);
    w.count(w.version == 5 ? bytes : (bytes > 0xffffffffull ? 0xffffffffull : bytes)); // vsize
    const uint64_t begin = w.b.size() + (w.version == 1 ? 4 : 8);
    w.offset(begin);
    File fl;
    fl.f = fopen(path, "wb");
    REQUIRE_IO(fl.f != nullptr, "cannot create %s", path);
    REQUIRE_IO(fwrite(w.b.data(), 1, w.b.size(), fl.f) == w.b.size(), "%s: write failed", path);
    std::vector<uint32_t> buf((size_t)1 << 20);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(color_czyx);
    for (uint64_t done = 0; done < n; done += buf.size()) {
        const size_t k = n - done < buf.size() ? (size_t)(n - done) : buf.size();
        for (size_t i = 0; i < k; ++i) buf[i] = bswap32(src[done + i]);
        REQUIRE_IO(fwrite(buf.data(), 4, k, fl.f) == k, "%s: write failed", path);
    }
    return VRM_OK;
    });
}

// loaders::netcdf::open_voxel_grid (netcdf.rs:34-69)
extern "C" int vrm_load_netcdf(vrm_ctx *ctx, const char *path)
{
    return vrm_guard("vrm_load_netcdf", VRM_ERR_IO, [&]() -> int {
    if (!ctx) {
        vrm_set_error("null context");
        return VRM_ERR_INVALID;
    }
    uint32_t x = 0, y = 0, z = 0;
    int rc = vrm_netcdf_read_color(path, &x, &y, &z, nullptr, 0);
    if (rc) return rc;
    // the header must not size an allocation the file cannot fill (deflate expands at most ~1032:1)
    struct stat st;
    REQUIRE_IO(stat(path, &st) == 0 && (double)x * y * z * 16.0 <= (double)st.st_size * 1040.0 + 65536.0,
               "%s: the header promises color(4,%u,%u,%u), more than its %lld bytes can hold", path, z, y, x, (long long)st.st_size);
    std::vector<float> data((size_t)4 * x * y * z);
    rc = vrm_netcdf_read_color(path, &x, &y, &z, data.data(), data.size());
    if (rc) return rc;
    rc = vrm_grid_resize(ctx, x, y, z);             // VoxelGrid::new (netcdf.rs:47)
    if (rc) return rc;
    return vrm_grid_upload_f32(ctx, data.data(), 4); // (v * 255.0) as u8 per channel (netcdf.rs:54-57), on the GPU
    });
}

// The NetCDF-4 (HDF5) container, what the reference's `netcdf::create` (netcdf.rs:10) makes: csrc/vrm_hdf5.cpp
extern "C" int vrm_netcdf_write_color_nc4(const char *path, uint32_t x, uint32_t y, uint32_t z, const float *color_czyx)
{
    return vrm_guard("vrm_netcdf_write_color_nc4", VRM_ERR_IO, [&]() -> int {
    if (!path || !color_czyx || x == 0 || y == 0 || z == 0) {
        vrm_set_error("null argument or empty volume");
        return VRM_ERR_INVALID;
    }
    return vrm_hdf5_write_color(path, x, y, z, color_czyx);
    });
}

// loaders::netcdf::write_voxel_grid (netcdf.rs:8-31)
static int write_voxel_grid(vrm_ctx *ctx, const char *path, bool nc4)
{
    if (!ctx) {
        vrm_set_error("null context");
        return VRM_ERR_INVALID;
    }
    const size_t n = ctx->vox_words;
    if (ctx->gdim[2] != ctx->dim[2]) {
        vrm_set_error("write_voxel_grid needs the whole volume in this context (slab-partitioned grid)");
        return VRM_ERR_STATE;
    }
    std::vector<uint8_t> vox(n * 4);
    int rc = vrm_grid_download_rgba8(ctx, vox.data(), vox.size());
    if (rc) return rc;
    std::vector<float> data(n * 4);
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < 4; ++c) data[(size_t)c * n + i] = (float)vox[4 * i + c] / 255.0f; // netcdf.rs:23
    return nc4 ? vrm_netcdf_write_color_nc4(path, ctx->dim[0], ctx->dim[1], ctx->dim[2], data.data())
               : vrm_netcdf_write_color(path, ctx->dim[0], ctx->dim[1], ctx->dim[2], data.data());
}

extern "C" int vrm_write_netcdf(vrm_ctx *ctx, const char *path)
{
    return vrm_guard("vrm_write_netcdf", VRM_ERR_IO, [&]() -> int { return write_voxel_grid(ctx, path, false); });
}

extern "C" int vrm_write_netcdf4(vrm_ctx *ctx, const char *path)
{
    return vrm_guard("vrm_write_netcdf4", VRM_ERR_IO, [&]() -> int { return write_voxel_grid(ctx, path, true); });
}
