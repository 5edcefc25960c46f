/* ORACLE (test infrastructure; never linked into or called by the product): plain-C restatement of
 *
 *   (1) OpenPCDet's iou3d_nms as vendored by the reference:
 *         opencood/pcdet_utils/iou3d_nms/src/iou3d_nms_kernel.cu  box_overlap :104-234, iou_bev :236-243 (as numbered
 *         in SURVEY.md: check_rect_cross :42-48, check_in_box2d :50-61 with MARGIN 1e-2, intersection :63-94,
 *         point_cmp :102-104), boxes_overlap_kernel / boxes_iou_bev_kernel, nms_kernel :267-311, iou_normal /
 *         nms_normal_kernel :314-372, and the host-side greedy reduction of the 64-bit masks iou3d_nms.cpp:116-135.
 *       All arithmetic is float32 in the reference's order of operations (compile with -ffp-contract=off).
 *       PARITY UNPINNED: the reference's own CPU twin (iou3d_cpu.cpp) includes <cuda.h> / <cuda_runtime_api.h>, which
 *       this ROCm image lacks, so it cannot be built here without stand-in headers; anchored by known-answer cases in
 *       tests/test_iou3d_voxel.py.
 *
 *   (2) spconv's point-to-voxel (spconv 1.2.1 VoxelGeneratorV2 / 2.x Point2VoxelCPU3d, the dependency behind
 *       opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:25-29, :54-68 -- not vendored, `pip install
 *       spconv-cu116`, README.md:116): points in input order; cell = floor((p - range_min) / voxel_size) per axis in
 *       float32, dropped when outside the grid; a cell seen for the first time becomes the next voxel unless
 *       max_voxels are in use (then the point is dropped); a point is appended to its voxel while it holds fewer than
 *       max_points. Coordinates are stored (z, y, x). PARITY UNPINNED (third-party arithmetic absent from
 *       /root/reference and from the image), restated from the published algorithm.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y; } Point;
static const float EPS = 1e-8f;

static float cross2(Point a, Point b) { return a.x * b.y - a.y * b.x; }
static float cross3(Point p1, Point p2, Point p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }
static int check_rect_cross(Point p1, Point p2, Point q1, Point q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}
static int check_in_box2d(const float* box, Point p) {
  const float MARGIN = 1e-2f;
  float center_x = box[0], center_y = box[1];
  float angle_cos = cosf(-box[6]), angle_sin = sinf(-box[6]);
  float rot_x = (p.x - center_x) * angle_cos + (p.y - center_y) * (-angle_sin);
  float rot_y = (p.x - center_x) * angle_sin + (p.y - center_y) * angle_cos;
  return fabsf(rot_x) < box[3] / 2 + MARGIN && fabsf(rot_y) < box[4] / 2 + MARGIN;
}
static int intersection(Point p1, Point p0, Point q1, Point q0, Point* ans) {
  if (check_rect_cross(p0, p1, q0, q1) == 0) return 0;
  float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > EPS) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}
static void rotate_around_center(Point center, float angle_cos, float angle_sin, Point* p) {
  float new_x = (p->x - center.x) * angle_cos + (p->y - center.y) * (-angle_sin) + center.x;
  float new_y = (p->x - center.x) * angle_sin + (p->y - center.y) * angle_cos + center.y;
  p->x = new_x; p->y = new_y;
}
static int point_cmp(Point a, Point b, Point center) {
  return atan2f(a.y - center.y, a.x - center.x) > atan2f(b.y - center.y, b.x - center.x);
}

float gc_oracle_box_overlap(const float* box_a, const float* box_b) {
  float a_angle = box_a[6], b_angle = box_b[6];
  float a_dx_half = box_a[3] / 2, b_dx_half = box_b[3] / 2, a_dy_half = box_a[4] / 2, b_dy_half = box_b[4] / 2;
  float a_x1 = box_a[0] - a_dx_half, a_y1 = box_a[1] - a_dy_half, a_x2 = box_a[0] + a_dx_half, a_y2 = box_a[1] + a_dy_half;
  float b_x1 = box_b[0] - b_dx_half, b_y1 = box_b[1] - b_dy_half, b_x2 = box_b[0] + b_dx_half, b_y2 = box_b[1] + b_dy_half;
  Point center_a = {box_a[0], box_a[1]}, center_b = {box_b[0], box_b[1]};
  Point ac[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0, 0}};
  Point bc[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0, 0}};
  float a_cos = cosf(a_angle), a_sin = sinf(a_angle), b_cos = cosf(b_angle), b_sin = sinf(b_angle);
  for (int k = 0; k < 4; k++) {
    rotate_around_center(center_a, a_cos, a_sin, &ac[k]);
    rotate_around_center(center_b, b_cos, b_sin, &bc[k]);
  }
  ac[4] = ac[0]; bc[4] = bc[0];
  Point cp[16], pc = {0, 0};
  int cnt = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      if (intersection(ac[i + 1], ac[i], bc[j + 1], bc[j], &cp[cnt])) {
        pc.x = pc.x + cp[cnt].x; pc.y = pc.y + cp[cnt].y;
        cnt++;
      }
  for (int k = 0; k < 4; k++) {
    if (check_in_box2d(box_a, bc[k])) { pc.x = pc.x + bc[k].x; pc.y = pc.y + bc[k].y; cp[cnt++] = bc[k]; }
    if (check_in_box2d(box_b, ac[k])) { pc.x = pc.x + ac[k].x; pc.y = pc.y + ac[k].y; cp[cnt++] = ac[k]; }
  }
  pc.x /= cnt; pc.y /= cnt;
  for (int j = 0; j < cnt - 1; j++)
    for (int i = 0; i < cnt - j - 1; i++)
      if (point_cmp(cp[i], cp[i + 1], pc)) { Point t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t; }
  float area = 0;
  for (int k = 0; k < cnt - 1; k++) {
    Point u = {cp[k].x - cp[0].x, cp[k].y - cp[0].y}, v = {cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y};
    area += cross2(u, v);
  }
  return fabsf(area) / 2.0f;
}
float gc_oracle_iou_bev(const float* a, const float* b) {
  float sa = a[3] * a[4], sb = b[3] * b[4], s = gc_oracle_box_overlap(a, b);
  return s / fmaxf(sa + sb - s, EPS);
}
float gc_oracle_iou_normal(const float* a, const float* b) {
  float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  float interS = width * height, Sa = a[3] * a[4], Sb = b[3] * b[4];
  return interS / fmaxf(Sa + Sb - interS, EPS);
}
/* mode 0: overlap area, 1: BEV IoU */
void gc_oracle_boxes_pairwise(const float* boxes_a, int num_a, const float* boxes_b, int num_b, int mode, float* out) {
  for (int i = 0; i < num_a; i++)
    for (int j = 0; j < num_b; j++)
      out[(size_t)i * num_b + j] = mode ? gc_oracle_iou_bev(boxes_a + i * 7, boxes_b + j * 7) : gc_oracle_box_overlap(boxes_a + i * 7, boxes_b + j * 7);
}
/* nms_kernel / nms_normal_kernel masks + iou3d_nms.cpp:116-135 greedy reduction; boxes already in score order. Returns num_to_keep. */
int gc_oracle_nms(const float* boxes, int n, float thresh, int normal, int64_t* keep) {
  const int B = 64, col_blocks = (n + B - 1) / B;
  uint64_t* mask = (uint64_t*)calloc((size_t)n * col_blocks + 1, sizeof(uint64_t));
  for (int i = 0; i < n; i++)
    for (int cb = 0; cb < col_blocks; cb++) {
      const int col_size = (n - cb * B) < B ? (n - cb * B) : B;
      uint64_t t = 0;
      const int start = (i / B == cb) ? (i % B) + 1 : 0;
      for (int k = start; k < col_size; k++) {
        const float* o = boxes + (size_t)(cb * B + k) * 7;
        const float v = normal ? gc_oracle_iou_normal(boxes + (size_t)i * 7, o) : gc_oracle_iou_bev(boxes + (size_t)i * 7, o);
        if (v > thresh) t |= 1ULL << k;
      }
      mask[(size_t)i * col_blocks + cb] = t;
    }
  uint64_t* remv = (uint64_t*)calloc(col_blocks + 1, sizeof(uint64_t));
  int num = 0;
  for (int i = 0; i < n; i++) {
    const int nblock = i / B, inblock = i % B;
    if (!(remv[nblock] & (1ULL << inblock))) {
      keep[num++] = i;
      for (int j = nblock; j < col_blocks; j++) remv[j] |= mask[(size_t)i * col_blocks + j];
    }
  }
  free(mask); free(remv);
  return num;
}

/* spconv point-to-voxel. points [n][nfeat] (x, y, z first); range6 = (x0, y0, z0, x1, y1, z1); outputs sized for max_voxels.
 * Returns the number of voxels. */
int gc_oracle_points_to_voxel(const float* points, int n, int nfeat, const float* voxel_size3, const float* range6,
                              int max_points, int max_voxels, float* voxels, int32_t* coords_zyx, int32_t* num_points) {
  int grid[3];
  for (int j = 0; j < 3; j++) grid[j] = (int)lrintf((range6[3 + j] - range6[j]) / voxel_size3[j]);
  const size_t cells = (size_t)grid[0] * grid[1] * grid[2];
  int32_t* lut = (int32_t*)malloc(cells * sizeof(int32_t));
  memset(lut, 0xff, cells * sizeof(int32_t));
  memset(voxels, 0, (size_t)max_voxels * max_points * nfeat * sizeof(float));
  memset(num_points, 0, (size_t)max_voxels * sizeof(int32_t));
  int voxel_num = 0;
  for (int i = 0; i < n; i++) {
    int c[3], failed = 0;
    for (int j = 0; j < 3; j++) {
      const float f = floorf((points[(size_t)i * nfeat + j] - range6[j]) / voxel_size3[j]);
      if (!(f >= 0.f && f < (float)grid[j])) { failed = 1; break; }
      c[j] = (int)f;
    }
    if (failed) continue;
    const size_t cell = ((size_t)c[2] * grid[1] + c[1]) * grid[0] + c[0];
    int v = lut[cell];
    if (v == -1) {
      if (voxel_num >= max_voxels) continue;
      v = voxel_num++;
      lut[cell] = v;
      coords_zyx[v * 3 + 0] = c[2]; coords_zyx[v * 3 + 1] = c[1]; coords_zyx[v * 3 + 2] = c[0];
    }
    const int k = num_points[v];
    if (k < max_points) {
      memcpy(voxels + ((size_t)v * max_points + k) * nfeat, points + (size_t)i * nfeat, nfeat * sizeof(float));
      num_points[v] = k + 1;
    }
  }
  free(lut);
  return voxel_num;
}
