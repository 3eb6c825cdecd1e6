/*
 * c_host.c -- the C ABI of libpcr_hip.so driven from plain C99 (no Python, no torch): what a
 * C/C++ host or an FFI binding in another language does.
 *
 *   c_host version                               print pcr_version() and the visible GPUs
 *   c_host run <kind> <target.f32> <scan.f32> [voxel_size]
 *       kind 0..3 = ICP / PlaneICP / VPlaneICP / NDT; the files are raw float32 (N,3) arrays.
 *       Prints one line "linearize" with the 29 sums at T = I (pcr.h: out[29]) and one line
 *       "align" with the iteration count and the 16 entries of the pose, all with %.17g.
 *
 * Build:  gcc -std=c99 -O2 -Iinclude examples/c_host.c -o c_host \
 *             -Lpoint_cloud_registration_amd -lpcr_hip -Wl,-rpath,$PWD/point_cloud_registration_amd
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pcr.h"

#define CHECK(expr)                                                              \
    do {                                                                         \
        pcr_status st_ = (expr);                                                 \
        if (st_ != PCR_OK) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #expr, st_, pcr_last_error());     \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static float *read_f32(const char *path, int64_t *n_points) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return NULL; }
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    float *buf = (float *)malloc(bytes > 0 ? (size_t)bytes : 4);
    if (fread(buf, 1, (size_t)bytes, f) != (size_t)bytes) { fclose(f); free(buf); return NULL; }
    fclose(f);
    *n_points = bytes / 12;
    return buf;
}

int main(int argc, char **argv) {
    if (argc >= 2 && strcmp(argv[1], "version") == 0) {
        int n = -1;
        CHECK(pcr_device_count(&n));
        printf("%s, %d GPU(s) visible\n", pcr_version(), n);
        return 0;
    }
    if (argc < 5 || strcmp(argv[1], "run") != 0) {
        fprintf(stderr, "usage: %s version | run <kind 0-3> <target.f32> <scan.f32> [voxel_size]\n", argv[0]);
        return 2;
    }
    const int kind = atoi(argv[2]);
    const double voxel_size = argc > 5 ? atof(argv[5]) : 1.0;
    const double max_dist = 2.0;
    int64_t nt = 0, ns = 0;
    float *target = read_f32(argv[3], &nt), *scan = read_f32(argv[4], &ns);
    if (!target || !scan) return 1;

    pcr_context *ctx = NULL;
    pcr_target *tgt = NULL;
    pcr_scan *sc = NULL;
    CHECK(pcr_context_create(0, &ctx));
    if (kind == PCR_ICP || kind == PCR_PLANE) {
        CHECK(pcr_target_points_create(ctx, target, nt, NULL, 0.0f, &tgt));
        if (kind == PCR_PLANE) CHECK(pcr_target_estimate_normals(tgt, 15, 1, NULL));
    } else {
        CHECK(pcr_target_voxels_create(ctx, target, 0, nt, voxel_size, 10, &tgt));
    }
    CHECK(pcr_scan_create(ctx, scan, ns, 0u, &sc));

    const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double out[29];
    CHECK(pcr_linearize(tgt, sc, kind, I4, max_dist, PCR_FLAG_ICP_RR_QUIRK, out));
    printf("linearize");
    for (int i = 0; i < 29; ++i) printf(" %.17g", out[i]);
    printf("\n");

    double T[16];
    int iters = 0;
    CHECK(pcr_align(tgt, sc, kind, I4, 30, 1e-3, max_dist, PCR_FLAG_ICP_RR_QUIRK, T, &iters, NULL));
    printf("align %d", iters);
    for (int i = 0; i < 16; ++i) printf(" %.17g", T[i]);
    printf("\n");

    CHECK(pcr_scan_destroy(sc));
    CHECK(pcr_target_destroy(tgt));
    CHECK(pcr_context_destroy(ctx));
    free(target);
    free(scan);
    return 0;
}
