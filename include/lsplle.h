/*
 * lsplle.h -- C ABI of the MI355X (gfx950) manifold projection of the APC audio features:
 * K nearest neighbours in the per-person feature database, then the locally-linear-embedding reconstruction of
 * every frame from its neighbours.  Exported by livespeechportraits_amd/liblspf2f.so.
 *
 * Reference path replaced (file:line under the reference tree), called at demo.py:196-200:
 *   funcs/utils.py:100-118   KNN_with_torch: |f|^2 + |b|^2 - 2 f.b^T, topk(K, largest=False)
 *   funcs/utils.py:121-158   solve_LLE_projection: w[1:] = solve(A^T A, A^T B), w[0] = 1 - sum, fuse = w . base
 *   funcs/utils.py:171-179   compute_LLE_projection_all_frame: Python loop over frames
 *   demo.py:200              audio_feats * (1 - LLE_percent) + feat_fuse * LLE_percent
 *
 * Stateless functions; conventions as lspf2f.h (0 / negative status, lsplle_last_error(), no device allocation,
 * asynchronous on `stream`).  All tensors are caller-owned device memory, fp32 unless noted.
 */
#ifndef LSPLLE_H
#define LSPLLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: exactly the functions declared below are exported */
#pragma GCC visibility push(default)

#define LSPLLE_OK 0
#define LSPLLE_ERR_INVALID_ARGUMENT (-1)
#define LSPLLE_ERR_UNSUPPORTED (-2)
#define LSPLLE_ERR_HIP (-4)
#define LSPLLE_ERR_SHAPE (-5)

#define LSPLLE_MAX_K 16

const char *lsplle_last_error(void);

/* Scratch for lsplle_knn: the [n][m] distance matrix and the row norms. */
size_t lsplle_knn_workspace_bytes(int n, int m);

/* KNN_with_torch (funcs/utils.py:100-118): ind[i][0..K) = database rows nearest to feats[i], nearest first.
 *   feats_dev [n][d], db_dev [m][d], ind_dev int64 [n][K];  d % 4 == 0, 1 <= K <= min(m, LSPLLE_MAX_K). */
int lsplle_knn(const float *feats_dev, int n, const float *db_dev, int m, int d, int K, int64_t *ind_dev,
               void *workspace_dev, size_t workspace_bytes, void *stream);

/* compute_LLE_projection_all_frame (funcs/utils.py:171-179) for all frames at once, optionally followed by the
 * blend of demo.py:200.
 *   ind_dev      int64 [n][K]   neighbour rows (ind[i][0] is the pivot f1 of the derivation in utils.py:126-141).
 *                               A row holding an index outside [0, m) -- lsplle_knn writes -1 when a feature row had fewer
 *                               than K finite distances (NaN input) -- reads nothing and produces NaN outputs for that row.
 *   weights_dev  double [n][K]  or NULL   (the reference's w is float64: utils.py:148)
 *   fuse_dev     float [n][d]   or NULL   feat_fuse
 *   blend_dev    float [n][d]   or NULL   feats * (1 - percent) + feat_fuse * percent
 * The (K-1)x(K-1) normal equations are solved in fp32 by LU with partial pivoting, as numpy.linalg.solve does for
 * float32 input; w[0] and the reconstruction are formed in double like the reference. */
int lsplle_solve(const float *feats_dev, int n, const float *db_dev, int m, int d, const int64_t *ind_dev, int K,
                 double *weights_dev, float *fuse_dev, float *blend_dev, float percent, void *stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* LSPLLE_H */
