/* lspraster.h -- C ABI of the landmark edge-map rasteriser of the render loop (second half).
 * Exported by livespeechportraits_amd/liblspf2f.so; gfx950 only, no CPU path.
 *
 * Replaces, per frame (reference file:line):
 *   datasets/face_dataset.py:276-281  get_data_test_mode: uint8 edge image -> float32 [1, H, W] / 255
 *   datasets/face_dataset.py:284-294  get_feature_image
 *   datasets/face_dataset.py:297-305  draw_shoulder_points   cv2.line(img, pt1, pt2, 255, 2) per chain edge
 *   datasets/face_dataset.py:311-322  draw_face_feature_maps cv2.line(...) per edge of part_list (:34-42)
 *   demo.py:262-265                   the host rasterisation + 1 MiB H2D copy per frame
 * The drawing primitive is OpenCV 4.4.0's cv::line for thickness > 1 (requirements.txt: opencv_python==4.4.0.40): a quad
 * filled by FillConvexPoly (16.16 fixed point, outline by the Line2 DDA) plus a filled Circle at both end points.  Every
 * primitive writes the same value, so the image is the UNION of the primitives' pixel sets and the edges can be rasterised
 * in parallel.  Parity is UNPINNED against cv2 (absent from the build image); the kernel is bit-exact to
 * oracle/raster_oracle.c, which restates the published algorithm.
 *
 * Conventions: device pointers, nothing allocated, no synchronisation, returns 0 or a negative code (lspraster_last_error()).
 */
#ifndef LSPRASTER_H
#define LSPRASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: exactly the functions declared below are exported */
#pragma GCC visibility push(default)

#define LSPRASTER_OK 0
#define LSPRASTER_ERR_INVALID_ARGUMENT (-1)
#define LSPRASTER_ERR_UNSUPPORTED (-2)
#define LSPRASTER_ERR_HIP (-3)

/* element type of the point array; floating-point coordinates are truncated toward zero on the device, which is what
 * `[int(flt) for flt in keypoints[i]]` (face_dataset.py:301-302, :318-319) does on the host */
#define LSPRASTER_POINTS_I32 0
#define LSPRASTER_POINTS_F32 1
#define LSPRASTER_POINTS_F64 2

#define LSPRASTER_MAX_THICKNESS 32
#define LSPRASTER_MAX_WIDTH 4096

/* B edge maps in one launch.
 *   points_dev    [batch][npoints][2] (x, y) of `point_dtype`
 *   segments_dev  int32 [nsegments][2]: indices (a, b) into a frame's points; shared by the batch (the topology --
 *                 part_list plus the two shoulder chains -- is constant); an index outside [0, npoints) skips the edge
 *   thickness     2..LSPRASTER_MAX_THICKNESS (the reference uses 2; thickness 1 is a different OpenCV routine)
 *   height,width  image size; width % 32 == 0, width <= LSPRASTER_MAX_WIDTH
 *   out_f32_dev   float32 [batch][1][height][width], exactly {0, 1}  (= uint8 {0,255} / 255.), or NULL
 *   out_u8_dev    uint8  [batch][height][width], {0, 255} (the reference's im_edges), or NULL -- at least one output */
int lspraster_edge_maps(const void *points_dev, int point_dtype, int batch, int npoints, const int32_t *segments_dev,
                        int nsegments, int thickness, int height, int width, float *out_f32_dev, unsigned char *out_u8_dev,
                        void *hip_stream);

const char *lspraster_last_error(void);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
