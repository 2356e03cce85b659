/* TEST INFRASTRUCTURE ONLY (oracle). Not linked into the product library.
 *
 * Plain-C restatement of the reference's tile hit map:
 *   find_block_covered_by_point      viewport_prediction/utils/common.py:37-43
 *   find_tiles_covered_by_viewport   viewport_prediction/utils/common.py:46-58
 *   _find_regions_covered_by_fov     viewport_prediction/utils/common.py:83-127
 *   pixel centre = (int(x*W), int(y*H)) viewport_prediction/utils/results.py:15,18 ; predict.py:40,43
 *   IoU = sum(a&b)/sum(a|b)          viewport_prediction/utils/results.py:21 ; predict.py:46
 *
 * A map is a uint64: bit (row*tile_num_width + col) set <=> viewport[row][col] == 1
 * (row-major flatten, as predict.py:41-42 `.reshape(-1)`).
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/liboracle_tilemap.so oracle/tilemap.c
 */
#include <stdint.h>
#include <math.h>

/* Python floor-mod for a positive modulus. */
static int pymod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }
/* Python floor-div for a positive divisor. */
static int pyfloordiv(int a, int b) { int q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) q--; return q; }

static void block_of_point(int x, int y, int bw, int bh, int *w, int *h) {
    *w = pyfloordiv(x, bw);
    *h = pyfloordiv(y, bh);
    if (x > 0 && x % bw == 0) *w -= 1;
    if (y > 0 && y % bh == 0) *h -= 1;
}

/* Python / numpy slice bound on an axis of length n */
static int slice_bound(int i, int n) { if (i < 0) { i += n; if (i < 0) i = 0; } else if (i > n) i = n; return i; }

typedef struct { int x1, y1, x2, y2; } region_t;

/* returns number of regions (0 if no case matches: the reference would raise UnboundLocalError) */
static int regions_of_fov(int x1, int y1, int x2, int y2, int W, int H, region_t *r) {
    if (x1 >= 0 && x2 <= W && y1 >= 0 && y2 <= H) { r[0] = (region_t){x1, y1, x2, y2}; return 1; }
    if (x1 < 0 && x2 <= W && y1 < 0 && y2 <= H) {            /* case 2 */
        r[0] = (region_t){0, 0, x2, y2};
        r[1] = (region_t){pymod(x1, W), 0, W, y2};
        r[2] = (region_t){0, pymod(y1, H), x2, H};
        r[3] = (region_t){pymod(x1, W), pymod(y1, H), W, H};
        return 4;
    }
    if (x1 >= 0 && x2 > W && y1 < 0 && y2 <= H) {            /* case 3 */
        r[0] = (region_t){0, 0, pymod(x2, W), y2};
        r[1] = (region_t){x1, 0, W, y2};
        r[2] = (region_t){0, pymod(y1, H), pymod(x2, W), H};
        r[3] = (region_t){x1, pymod(y1, H), W, H};
        return 4;
    }
    if (x1 < 0 && x2 <= W && y1 >= 0 && y2 > H) {            /* case 4 */
        r[0] = (region_t){0, 0, x2, pymod(y2, H)};
        r[1] = (region_t){pymod(x1, W), 0, W, pymod(y2, H)};
        r[2] = (region_t){0, y1, x2, H};
        r[3] = (region_t){pymod(x1, W), y1, W, H};
        return 4;
    }
    if (x1 >= 0 && x2 > W && y1 >= 0 && y2 > H) {            /* case 5 */
        r[0] = (region_t){0, 0, pymod(x2, W), pymod(y2, H)};
        r[1] = (region_t){x1, 0, W, pymod(y2, H)};
        r[2] = (region_t){0, y1, pymod(x2, W), H};
        r[3] = (region_t){x1, y1, W, H};
        return 4;
    }
    if (x1 < 0 && x2 <= W && y1 >= 0 && y2 <= H) {           /* case 6 */
        r[0] = (region_t){0, y1, x2, y2};
        r[1] = (region_t){pymod(x1, W), y1, W, y2};
        return 2;
    }
    if (x1 >= 0 && x2 > W && y1 >= 0 && y2 <= H) {           /* case 7 */
        r[0] = (region_t){0, y1, pymod(x2, W), y2};
        r[1] = (region_t){x1, y1, W, y2};
        return 2;
    }
    if (x1 >= 0 && x2 <= W && y1 < 0 && y2 <= H) {           /* case 8 */
        r[0] = (region_t){x1, 0, x2, y2};
        r[1] = (region_t){x1, pymod(y1, H), x2, H};
        return 2;
    }
    if (x1 >= 0 && x2 <= W && y1 >= 0 && y2 > H) {           /* case 9 */
        r[0] = (region_t){x1, 0, x2, pymod(y2, H)};
        r[1] = (region_t){x1, y1, x2, H};
        return 2;
    }
    return 0;
}

uint64_t oracle_tilemap_px(int x, int y, int W, int H, int tw, int th, int nw, int nh, int fov_w, int fov_h) {
    int hw = fov_w / 2, hh = fov_h / 2;
    region_t r[4];
    int n = regions_of_fov(x - hw, y - hh, x + hw, y + hh, W, H, r);
    uint64_t m = 0;
    for (int k = 0; k < n; ++k) {
        int tx1, ty1, tx2, ty2;
        block_of_point(r[k].x1, r[k].y1, tw, th, &tx1, &ty1);
        block_of_point(r[k].x2, r[k].y2, tw, th, &tx2, &ty2);
        /* numpy slice semantics viewport[ty1:ty2+1, tx1:tx2+1] = 1: a negative bound counts from the END of the axis (only
         * then is it clipped to 0), a bound past the end is clipped to the length, a reversed range is empty.  Negative bounds
         * arise for viewport centres left of / above the frame by more than half a FoV (raw model output; the linear-regression
         * baseline extrapolating across a wrap-around jump) -- pinned by tests/golden/tilemap_px_outside.npz. */
        int ya = slice_bound(ty1, nh), yb = slice_bound(ty2 + 1, nh);
        int xa = slice_bound(tx1, nw), xb = slice_bound(tx2 + 1, nw);
        for (int yy = ya; yy < yb; ++yy)
            for (int xx = xa; xx < xb; ++xx) m |= (uint64_t)1 << (yy * nw + xx);
    }
    return m;
}

/* xy: n x 2 float32 normalised coordinates; pixel = (int)(v * size) with the multiply done in
 * float64 exactly like Python's `int(np.float32 * int)` under numpy 2.x?  NO: numpy-2 keeps
 * float32 for np.float32 * python-int (NEP 50), so the product is rounded to float32 first. */
void oracle_tilemap_xy(const float *xy, int n, int W, int H, int nw, int nh, int fov_w, int fov_h, uint64_t *out) {
    int tw = W / nw, th = H / nh;
    for (int i = 0; i < n; ++i) {
        float fx = xy[2 * i] * (float)W, fy = xy[2 * i + 1] * (float)H;
        out[i] = oracle_tilemap_px((int)fx, (int)fy, W, H, tw, th, nw, nh, fov_w, fov_h);
    }
}

static int popc(uint64_t v) { int c = 0; while (v) { v &= v - 1; ++c; } return c; }

void oracle_tilemap_iou(const uint64_t *a, const uint64_t *b, int n, double *iou) {
    for (int i = 0; i < n; ++i) iou[i] = (double)popc(a[i] & b[i]) / (double)popc(a[i] | b[i]);
}
