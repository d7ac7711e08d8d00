/* voxel_port.c -- TEST INFRASTRUCTURE ONLY (oracle).  CPU restatement of the reference's voxel containers,
 * Planner/src/voxel_map.cpp:5-76 (voxel_map<Cont>::add_point_cloud / add_point / get_voxel_cloud and
 * voxel_value_map::add_point), processed strictly in input order like the std::set / std::map originals.
 *
 * PARITY UNPINNED: voxel_map.cpp needs PCL and Eigen headers (absent here), and the reference holds no fixture for it;
 * this file follows the cited lines.  tests/test_voxel.py additionally checks it against an independent numpy
 * formulation (np.round is half-to-even, so that check builds round-half-away itself).
 *
 *   (int) round(point / res)          voxel_map.cpp:5-16   (float coordinates are widened: float / double -> double)
 *   set.insert(...).second            :29, :39             (new voxel -> append the centre)
 *   emplace_back(x*res, y*res, z*res) :30, :40, :68        (int * double; narrowed to float by PointXYZ / Vector3f)
 *   map.insert({key, map.size()})     :66                  (voxel_value_map: value = number of voxels before the insert)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double res;
    int64_t size, cap;        /* voxels */
    int32_t *key;             /* [cap][3] in first-seen order */
    int64_t tcap;             /* hash table (power of two), entries = voxel id or -1 */
    int64_t *table;
} ovox;

static uint64_t ovox_hash(int32_t x, int32_t y, int32_t z)
{
    uint64_t h = (uint64_t)(uint32_t)x * 0x9E3779B97F4A7C15ull;
    h ^= ((uint64_t)(uint32_t)y + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
    h ^= ((uint64_t)(uint32_t)z + 0x165667B1ull) * 0xD6E8FEB86659FD93ull;
    h ^= h >> 29;
    return h;
}

ovox *ovox_create(double res)
{
    ovox *m = (ovox *)calloc(1, sizeof *m);
    m->res = res;
    m->cap = 1024;
    m->key = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)m->cap);
    m->tcap = 4096;
    m->table = (int64_t *)malloc(sizeof(int64_t) * (size_t)m->tcap);
    for (int64_t i = 0; i < m->tcap; i++) m->table[i] = -1;
    return m;
}

void ovox_destroy(ovox *m)
{
    if (!m) return;
    free(m->key);
    free(m->table);
    free(m);
}

int64_t ovox_size(const ovox *m) { return m->size; }

static void ovox_grow_table(ovox *m)
{
    const int64_t ncap = m->tcap * 2;
    int64_t *t = (int64_t *)malloc(sizeof(int64_t) * (size_t)ncap);
    for (int64_t i = 0; i < ncap; i++) t[i] = -1;
    for (int64_t id = 0; id < m->size; id++) {
        const int32_t *k = m->key + 3 * id;
        uint64_t s = ovox_hash(k[0], k[1], k[2]) & (uint64_t)(ncap - 1);
        while (t[s] >= 0) s = (s + 1) & (uint64_t)(ncap - 1);
        t[s] = id;
    }
    free(m->table);
    m->table = t;
    m->tcap = ncap;
}

/* returns the voxel id; *added = 1 when the voxel is new */
static int64_t ovox_insert(ovox *m, int32_t x, int32_t y, int32_t z, int *added)
{
    uint64_t s = ovox_hash(x, y, z) & (uint64_t)(m->tcap - 1);
    while (m->table[s] >= 0) {
        const int32_t *k = m->key + 3 * m->table[s];
        if (k[0] == x && k[1] == y && k[2] == z) { *added = 0; return m->table[s]; }
        s = (s + 1) & (uint64_t)(m->tcap - 1);
    }
    if (m->size == m->cap) {
        m->cap *= 2;
        m->key = (int32_t *)realloc(m->key, sizeof(int32_t) * 3 * (size_t)m->cap);
    }
    const int64_t id = m->size++;
    m->key[3 * id] = x; m->key[3 * id + 1] = y; m->key[3 * id + 2] = z;
    m->table[s] = id;
    *added = 1;
    if (m->size * 2 > m->tcap) ovox_grow_table(m);
    return id;
}

/* add_point_cloud over n records (stride in bytes; float or double coordinates); optional per-point outputs */
int64_t ovox_add(ovox *m, const void *pts, int64_t n, int64_t stride_bytes, int is_f64, uint8_t *is_new, int32_t *voxel_index)
{
    int64_t n_new = 0;
    const unsigned char *p = (const unsigned char *)pts;
    for (int64_t i = 0; i < n; i++, p += stride_bytes) {
        double c[3];
        if (is_f64) memcpy(c, p, sizeof c);
        else { float f[3]; memcpy(f, p, sizeof f); c[0] = f[0]; c[1] = f[1]; c[2] = f[2]; }
        const int32_t x = (int32_t)round(c[0] / m->res), y = (int32_t)round(c[1] / m->res), z = (int32_t)round(c[2] / m->res);
        int added;
        const int64_t id = ovox_insert(m, x, y, z, &added);
        n_new += added;
        if (is_new) is_new[i] = (uint8_t)added;
        if (voxel_index) voxel_index[i] = (int32_t)id;
    }
    return n_new;
}

void ovox_get_keys(const ovox *m, int32_t *out) { memcpy(out, m->key, sizeof(int32_t) * 3 * (size_t)m->size); }

void ovox_get_f64(const ovox *m, double *out)
{
    for (int64_t i = 0; i < 3 * m->size; i++) out[i] = m->key[i] * m->res;
}

void ovox_get_f32(const ovox *m, float *out)
{
    for (int64_t i = 0; i < 3 * m->size; i++) out[i] = (float)(m->key[i] * m->res);
}
