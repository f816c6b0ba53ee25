/* trace_obj.c -- the whole path through the C-ABI from plain C, no C++ and no Python:
 * OBJ file -> flat scene -> accel on the GPU -> one closest-hit batch of pinhole camera rays ->
 * normals on the device -> a shaded float framebuffer -> BMP.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/trace_obj.c -o trace_obj -Lyart_amd -lyart_amd -Wl,-rpath,$PWD/yart_amd -lm
 *   ./trace_obj tests/golden/obj/ terrain.obj out.bmp [width height]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "yart_amd.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int st_ = (call);                                                                  \
        if (st_ != YART_OK) {                                                              \
            fprintf(stderr, "%s: %s: %s\n", #call, yart_status_string(st_), yart_last_error()); \
            return 3;                                                                      \
        }                                                                                  \
    } while (0)

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: trace_obj <folder/> <file.obj> <out.bmp> [width height]\n"); return 2; }
    const int width = argc > 5 ? atoi(argv[4]) : 640, height = argc > 5 ? atoi(argv[5]) : 360;

    YartObjScene* obj = NULL;
    CHECK(yart_obj_load(argv[1], argv[2], &obj));
    const size_t nv = yart_obj_num_vertices(obj), nt = yart_obj_num_triangles(obj), ns = yart_obj_num_spheres(obj);
    float* verts = malloc(nv * 12 + 4), *normals = malloc(nv * 12 + 4), *spheres = malloc(ns * 16 + 4);
    int32_t* tris = malloc(nt * 20 + 4);
    CHECK(yart_obj_get(obj, verts, normals, tris, spheres, NULL, NULL));
    yart_obj_free(obj);

    float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
    for (size_t i = 0; i < nv * 3; ++i) {
        if (!(fabsf(verts[i]) < 1e30f)) continue;
        if (verts[i] < lo[i % 3]) lo[i % 3] = verts[i];
        if (verts[i] > hi[i % 3]) hi[i % 3] = verts[i];
    }
    const float c[3] = { (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, (lo[2] + hi[2]) / 2 };
    const float e[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] };
    const float eye[3] = { c[0] - 0.9f * e[0], c[1] + 1.2f * e[1] + 0.3f * e[0], c[2] - 0.2f * e[2] };

    YartAccel* accel = NULL;
    CHECK(yart_accel_create(verts, nv, tris, nt, spheres, ns, 0, &accel));
    CHECK(yart_accel_set_surface_data(accel, verts, nv, tris, nt, normals, spheres, ns));

    /* pinhole camera looking at the centre of the scene */
    const size_t n = (size_t)width * height;
    YartRay* rays = malloc(n * sizeof *rays);
    YartHit* hits = malloc(n * sizeof *hits);
    YartSurface* surf = malloc(n * sizeof *surf);
    float f[3] = { c[0] - eye[0], c[1] - eye[1], c[2] - eye[2] };
    float fl = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (int a = 0; a < 3; ++a) f[a] /= fl;
    float r[3] = { -f[2], 0.0f, f[0] };                                 /* cross(f, up = +y), not yet unit */
    float rl = sqrtf(r[0] * r[0] + r[2] * r[2]);
    for (int a = 0; a < 3; ++a) r[a] /= rl;
    const float u[3] = { r[1] * f[2] - r[2] * f[1], r[2] * f[0] - r[0] * f[2], r[0] * f[1] - r[1] * f[0] };
    const float tanHalf = 0.45f;
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            const float sx = ((x + 0.5f) / width * 2 - 1) * tanHalf * width / height, sy = (1 - (y + 0.5f) / height * 2) * tanHalf;
            float d[3];
            for (int a = 0; a < 3; ++a) d[a] = f[a] + sx * r[a] + sy * u[a];
            const float dl = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            YartRay* q = &rays[(size_t)y * width + x];
            q->ox = eye[0]; q->oy = eye[1]; q->oz = eye[2]; q->tmin = 0.0f;
            q->dx = d[0] / dl; q->dy = d[1] / dl; q->dz = d[2] / dl; q->tmax = INFINITY;
        }
    }
    CHECK(yart_intersect_closest(accel, rays, n, hits, YART_DEFAULT));
    CHECK(yart_surface(accel, rays, hits, n, surf));

    /* headlight shading: |normal . direction| */
    float* rgb = malloc(n * 12);
    size_t nHit = 0;
    for (size_t i = 0; i < n; ++i) {
        float v = 0.0f;
        if (surf[i].prim != -1) {
            const float* nn = surf[i].shading_normal;
            const float len = sqrtf(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
            v = fabsf(nn[0] * rays[i].dx + nn[1] * rays[i].dy + nn[2] * rays[i].dz) / (len > 0 ? len : 1.0f);
            ++nHit;
        }
        rgb[i * 3] = v * 1.5f; rgb[i * 3 + 1] = v * 1.2f; rgb[i * 3 + 2] = v;
    }
    CHECK(yart_film_write_bmp(argv[3], rgb, width, height));
    printf("%zu triangles, %zu spheres, %zu rays, %zu hits -> %s\n", nt, ns, n, nHit, argv[3]);
    yart_accel_destroy(accel);
    free(verts); free(normals); free(spheres); free(tris); free(rays); free(hits); free(surf); free(rgb);
    return 0;
}
