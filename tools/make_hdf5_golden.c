/* Golden NetCDF-4 (= HDF5) files for tests/test_hdf5_golden_cpu.py, written by the REAL HDF5 library (libhdf5 1.10.6, the one this
 * image carries under /opt/conda; netcdf-c, which the reference's `netcdf` crate wraps, makes exactly these calls: H5Dcreate of the
 * variable, one dimension-scale dataset per dimension, H5DSset_scale / H5DSattach_scale, the `_Netcdf4Dimid` and `_NCProperties`
 * attributes).  They pin csrc/vrm_hdf5.cpp -- reader and writer -- against an implementation that is not ours.
 *
 *   gcc -O1 -o /tmp/make_hdf5_golden tools/make_hdf5_golden.c -I/opt/conda/include -L/opt/conda/lib -Wl,-rpath,/opt/conda/lib -lhdf5_hl -lhdf5
 *   /tmp/make_hdf5_golden tests/golden/hdf5
 *
 * The value of element i (row-major over c,z,y,x) is ((uint32_t)(i * 2654435761u) >> 8) / 2^24: exact in f32, restated by the test.
 * Files (all hold `color(c=4,z,y,x)` f32 little-endian unless said otherwise):
 *   lib_earliest_contiguous.nc   z,y,x = 3,5,6   format bounds (earliest, v18): superblock 0, version-1 object headers, symbol table
 *   lib_earliest_chunked.nc      z,y,x = 5,6,7   chunks of (1,2,4,3) -- ragged edges in every axis --, no filter
 *   lib_v18_deflate_shuffle.nc   z,y,x = 7,10,13 bounds (v18, v18): superblock 2, OHDR headers, link messages; shuffle + deflate 4
 *   lib_v18_fletcher_bigendian.nc z,y,x = 4,4,6  big-endian f32, chunks (4,2,2,3), fletcher32
 *   lib_v18_compact.nc           z,y,x = 2,3,4   compact layout (data inside the object header)
 *   lib_latest_chunked.nc        z,y,x = 6,5,9   bounds (latest, latest) of 1.10: superblock 3, layout message version 4 with a fixed-array
 *                                                chunk index, shuffle + deflate (netcdf-c stays below this format; h5py's libver="latest" writes it)
 *   lib_latest_paged.nc          z,y,x = 10,10,9 the same with 1200 chunks of (1,1,1,3): the fixed array's entries come in pages of 1024
 *   lib_latest_implicit.nc       z,y,x = 4,5,6   latest bounds, no filter, space allocated early: the implicit chunk index
 *   lib_latest_single_chunk.nc   z,y,x = 3,4,5   latest bounds, one chunk of the dataset's size, deflate
 *   lib_fill_value.nc            z,y,x = 4,6,8   chunks (4,2,3,4), fill value 9.96921e+36 (netCDF's NC_FILL_FLOAT) defined, and only the hyperslab
 *                                                z < 2 written: the other chunks do not exist and read as the fill value
 *   lib_f64_shuffle_deflate.nc   z,y,x = 5,4,7   `color` stored as little-endian binary64 (values i*2654435761 mod 2^32 / 2^32, which binary32 cannot hold:
 *                                                the reader hands them over as (float)value, like nc_get_vara_float), chunks (2,2,3,4), shuffle over 8 bytes
 *   lib_i16_bigendian.nc         z,y,x = 3,3,5   `color` as big-endian signed 16-bit integers (the hash's top 16 bits), contiguous
 *   lib_u8_chunked.nc            z,y,x = 4,4,4   `color` as unsigned bytes (the hash's top 8 bits), chunks (2,4,2,4)
 *   lib_many_attributes.nc       z,y,x = 3,4,5   v18 bounds, 12 attributes on `color`: dense attribute storage (fractal heap + B-tree v2)
 *   lib_many_variables.nc        z,y,x = 3,4,5   v18 bounds, 10 more datasets in the root group: dense link storage
 *   lib_hundred_variables.nc     z,y,x = 2,3,4   120 more: the link heap's root is an indirect block, the name index a B-tree of depth 1
 */
#include <hdf5.h>
#include <hdf5_hl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { if ((x) < 0) { fprintf(stderr, "%s:%d: %s failed\n", __FILE__, __LINE__, #x); exit(1); } } while (0)

static float *values(size_t n)
{
    float *v = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) v[i] = (float)(((uint32_t)i * 2654435761u) >> 8) / 16777216.0f;
    return v;
}

static void attr_i32(hid_t obj, const char *name, int value)
{
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(obj, name, H5T_NATIVE_INT, sp, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(a);
    CHECK(H5Awrite(a, H5T_NATIVE_INT, &value));
    H5Aclose(a);
    H5Sclose(sp);
}

static void attr_str(hid_t obj, const char *name, const char *text)
{
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, strlen(text) + 1);
    hid_t a = H5Acreate2(obj, name, t, sp, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(a);
    CHECK(H5Awrite(a, t, text));
    H5Aclose(a);
    H5Tclose(t);
    H5Sclose(sp);
}

enum { CONTIGUOUS, CHUNKED, COMPACT };
enum { AS_F32, AS_F64, AS_I16BE, AS_U8 };
static int stored_as = AS_F32; /* the file type of `color` in the next make() */

enum { MANY_ATTRIBUTES = 1, MANY_VARIABLES = 2, ALLOC_EARLY = 4, PARTIAL_WITH_FILL = 8, HUNDRED_VARIABLES = 16 };

static void make(const char *dir, const char *name, hsize_t z, hsize_t y, hsize_t x, H5F_libver_t lo, H5F_libver_t hi, int layout,
                 const hsize_t *chunk, int shuffle, int deflate, int fletcher, int big_endian, int extras)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    hid_t fapl = H5Pcreate(H5P_FILE_ACCESS);
    CHECK(H5Pset_libver_bounds(fapl, lo, hi));
    hid_t fcpl = H5Pcreate(H5P_FILE_CREATE);
    /* (netcdf-c tracks and indexes creation order of links and attributes) */
    CHECK(H5Pset_link_creation_order(fcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED));
    CHECK(H5Pset_attr_creation_order(fcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED));
    hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, fcpl, fapl);
    CHECK(f);
    attr_str(f, "_NCProperties", "version=2,netcdf=4.7.4,hdf5=1.10.6");

    const char *dn[4] = {"c", "z", "y", "x"};
    const hsize_t dl[4] = {4, z, y, x};
    hid_t scale[4];
    for (int i = 0; i < 4; ++i) {
        hid_t sp = H5Screate_simple(1, &dl[i], NULL);
        hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
        CHECK(H5Pset_attr_creation_order(dcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED));
        scale[i] = H5Dcreate2(f, dn[i], H5T_IEEE_F32BE, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT); /* (a dimension without a variable: what netcdf-c makes) */
        CHECK(scale[i]);
        char label[96];
        snprintf(label, sizeof label, "This is a netCDF dimension but not a netCDF variable.%10d", (int)dl[i]);
        CHECK(H5DSset_scale(scale[i], label));
        attr_i32(scale[i], "_Netcdf4Dimid", i);
        H5Pclose(dcpl);
        H5Sclose(sp);
    }

    hid_t sp = H5Screate_simple(4, dl, NULL);
    hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
    CHECK(H5Pset_attr_creation_order(dcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED));
    if (layout == CHUNKED) {
        CHECK(H5Pset_chunk(dcpl, 4, chunk));
        if (shuffle) CHECK(H5Pset_shuffle(dcpl));
        if (deflate) CHECK(H5Pset_deflate(dcpl, (unsigned)deflate));
        if (fletcher) CHECK(H5Pset_fletcher32(dcpl));
    } else if (layout == COMPACT) {
        CHECK(H5Pset_layout(dcpl, H5D_COMPACT));
    }
    if (extras & ALLOC_EARLY) CHECK(H5Pset_alloc_time(dcpl, H5D_ALLOC_TIME_EARLY));
    if (extras & PARTIAL_WITH_FILL) {
        const float nc_fill = 9.9692099683868690e+36f;
        CHECK(H5Pset_fill_value(dcpl, H5T_NATIVE_FLOAT, &nc_fill));
    }
    hid_t ftype = stored_as == AS_F64 ? H5T_IEEE_F64LE : stored_as == AS_I16BE ? H5T_STD_I16BE : stored_as == AS_U8 ? H5T_STD_U8LE
                : big_endian ? H5T_IEEE_F32BE : H5T_IEEE_F32LE;
    hid_t color = H5Dcreate2(f, "color", ftype, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT);
    CHECK(color);
    const size_t n = (size_t)(4 * z * y * x);
    float *v = values(n);
    if (stored_as != AS_F32) {
        /* written from memory of the file's own type: no conversion on the way in */
        void *w = malloc(n * 8);
        for (size_t i = 0; i < n; ++i) {
            const uint32_t hsh = (uint32_t)i * 2654435761u;
            if (stored_as == AS_F64) ((double *)w)[i] = (double)hsh / 4294967296.0;
            else if (stored_as == AS_I16BE) ((int16_t *)w)[i] = (int16_t)(hsh >> 16);
            else ((uint8_t *)w)[i] = (uint8_t)(hsh >> 24);
        }
        CHECK(H5Dwrite(color, stored_as == AS_F64 ? H5T_NATIVE_DOUBLE : stored_as == AS_I16BE ? H5T_NATIVE_INT16 : H5T_NATIVE_UINT8, H5S_ALL, H5S_ALL,
                       H5P_DEFAULT, w));
        free(w);
    } else
    if (extras & PARTIAL_WITH_FILL) {
        /* z < 2 only; the values are those of the full array at the same places */
        const hsize_t start[4] = {0, 0, 0, 0}, count[4] = {4, 2, y, x};
        hid_t fs = H5Dget_space(color), ms = H5Screate_simple(4, dl, NULL);
        CHECK(H5Sselect_hyperslab(fs, H5S_SELECT_SET, start, NULL, count, NULL));
        CHECK(H5Sselect_hyperslab(ms, H5S_SELECT_SET, start, NULL, count, NULL));
        CHECK(H5Dwrite(color, H5T_NATIVE_FLOAT, ms, fs, H5P_DEFAULT, v));
        H5Sclose(ms);
        H5Sclose(fs);
    } else {
        CHECK(H5Dwrite(color, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, v));
    }
    free(v);
    for (unsigned i = 0; i < 4; ++i) CHECK(H5DSattach_scale(color, scale[i], i));
    if (extras & (MANY_ATTRIBUTES | MANY_VARIABLES)) {
        char an[32];
        for (int i = 0; i < 12 && (extras & MANY_ATTRIBUTES); ++i) {
            snprintf(an, sizeof an, "note_%02d", i);
            attr_str(color, an, "an attribute the reader has no use for, one of enough to leave the object header");
        }
        for (int i = 0; i < ((extras & HUNDRED_VARIABLES) ? 120 : 10) && (extras & MANY_VARIABLES); ++i) {
            snprintf(an, sizeof an, "other_variable_%03d", (i * 37) % 120); /* (not in name order) */
            hsize_t one = 3;
            hid_t s1 = H5Screate_simple(1, &one, NULL);
            hid_t d1 = H5Dcreate2(f, an, H5T_NATIVE_INT, s1, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            CHECK(d1);
            const int w[3] = {i, i + 1, i + 2};
            CHECK(H5Dwrite(d1, H5T_NATIVE_INT, H5S_ALL, H5S_ALL, H5P_DEFAULT, w));
            H5Dclose(d1);
            H5Sclose(s1);
        }
    }
    H5Dclose(color);
    for (int i = 0; i < 4; ++i) H5Dclose(scale[i]);
    H5Pclose(dcpl);
    H5Sclose(sp);
    H5Pclose(fcpl);
    H5Pclose(fapl);
    CHECK(H5Fclose(f));
    printf("%s\n", path);
}

int main(int argc, char **argv)
{
    if (argc != 2) {
        fprintf(stderr, "usage: %s <output directory>\n", argv[0]);
        return 2;
    }
    const char *d = argv[1];
    const hsize_t c1[4] = {1, 2, 4, 3}, c2[4] = {2, 3, 4, 5}, c3[4] = {4, 2, 2, 3}, c4[4] = {1, 3, 2, 4};
    make(d, "lib_earliest_contiguous.nc", 3, 5, 6, H5F_LIBVER_EARLIEST, H5F_LIBVER_V18, CONTIGUOUS, NULL, 0, 0, 0, 0, 0);
    make(d, "lib_earliest_chunked.nc", 5, 6, 7, H5F_LIBVER_EARLIEST, H5F_LIBVER_V18, CHUNKED, c1, 0, 0, 0, 0, 0);
    make(d, "lib_v18_deflate_shuffle.nc", 7, 10, 13, H5F_LIBVER_V18, H5F_LIBVER_V18, CHUNKED, c2, 1, 4, 0, 0, 0);
    make(d, "lib_v18_fletcher_bigendian.nc", 4, 4, 6, H5F_LIBVER_V18, H5F_LIBVER_V18, CHUNKED, c3, 0, 0, 1, 1, 0);
    make(d, "lib_v18_compact.nc", 2, 3, 4, H5F_LIBVER_V18, H5F_LIBVER_V18, COMPACT, NULL, 0, 0, 0, 0, 0);
    make(d, "lib_latest_chunked.nc", 6, 5, 9, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST, CHUNKED, c4, 1, 2, 0, 0, 0);
    const hsize_t c5[4] = {1, 1, 1, 3}, c6[4] = {2, 2, 5, 3}, c7[4] = {4, 3, 4, 5}, c8[4] = {4, 2, 3, 4};
    make(d, "lib_latest_paged.nc", 10, 10, 9, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST, CHUNKED, c5, 0, 0, 0, 0, 0);
    make(d, "lib_latest_implicit.nc", 4, 5, 6, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST, CHUNKED, c6, 0, 0, 0, 0, ALLOC_EARLY);
    make(d, "lib_latest_single_chunk.nc", 3, 4, 5, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST, CHUNKED, c7, 0, 3, 0, 0, 0);
    make(d, "lib_fill_value.nc", 4, 6, 8, H5F_LIBVER_EARLIEST, H5F_LIBVER_V18, CHUNKED, c8, 0, 0, 0, 0, PARTIAL_WITH_FILL);
    const hsize_t c9[4] = {2, 2, 3, 4}, c10[4] = {2, 4, 2, 4};
    stored_as = AS_F64;
    make(d, "lib_f64_shuffle_deflate.nc", 5, 4, 7, H5F_LIBVER_EARLIEST, H5F_LIBVER_V18, CHUNKED, c9, 1, 5, 0, 0, 0);
    stored_as = AS_I16BE;
    make(d, "lib_i16_bigendian.nc", 3, 3, 5, H5F_LIBVER_V18, H5F_LIBVER_V18, CONTIGUOUS, NULL, 0, 0, 0, 0, 0);
    stored_as = AS_U8;
    make(d, "lib_u8_chunked.nc", 4, 4, 4, H5F_LIBVER_EARLIEST, H5F_LIBVER_V18, CHUNKED, c10, 0, 0, 0, 0, 0);
    stored_as = AS_F32;
    make(d, "lib_many_attributes.nc", 3, 4, 5, H5F_LIBVER_V18, H5F_LIBVER_V18, CONTIGUOUS, NULL, 0, 0, 0, 0, MANY_ATTRIBUTES);
    make(d, "lib_many_variables.nc", 3, 4, 5, H5F_LIBVER_V18, H5F_LIBVER_V18, CONTIGUOUS, NULL, 0, 0, 0, 0, MANY_VARIABLES);
    make(d, "lib_hundred_variables.nc", 2, 3, 4, H5F_LIBVER_V18, H5F_LIBVER_V18, CONTIGUOUS, NULL, 0, 0, 0, 0, MANY_VARIABLES | HUNDRED_VARIABLES);
    return 0;
}
