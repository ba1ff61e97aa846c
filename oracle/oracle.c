/* CPU oracle for the multiview fusion hot path -- plain C restatement.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under mvdetr_amd/ may link, load or call this library; only
 * tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py do, as the checker / the
 * timed CPU baseline (through oracle/c_oracle.py).
 *
 * What it restates (reference files under /root/reference, read-only):
 *   oracle_msda_forward_{f32,f64}    ms_deform_attn_core_pytorch
 *                                    multiview_detector/models/ops/functions/ms_deform_attn_func.py:41-61
 *   oracle_msda_backward_{f32,f64}   autograd of the above == ms_deform_attn_col2im_bilinear
 *                                    multiview_detector/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159
 *   oracle_warp_perspective*_{..}    kornia.warp_perspective as called at
 *                                    multiview_detector/models/mvdetr.py:194-195
 *
 * Parity status: the MSDA functions are PINNED against golden vectors produced by the reference's
 * own Python code (tests/golden/make_golden.py -> tests/golden/msda_*.npz; tests/test_oracle.py).
 * The reference's native sources are CUDA-only (THC headers, <<<>>>; setup.py refuses to build
 * without CUDA), i.e. unbuildable here, so there is no oracle/_ref.
 * The warp is PARITY UNPINNED: kornia (un-pinned dependency, README.md:42; 0.5.x era) is neither
 * vendored nor installed, and the reference holds no test vector for that call; this file restates
 * kornia 0.5's published algorithm and is cross-checked only against oracle/torch_oracle.py,
 * whose sampler half is torch's own F.grid_sample.
 */
#include <math.h>
#include <stdint.h>

#define T float
#define FN(x) x##_f32
#define FLOOR floorf
#define FABS fabsf
#include "oracle_impl.h"
#undef T
#undef FN
#undef FLOOR
#undef FABS

#define T double
#define FN(x) x##_f64
#define FLOOR floor
#define FABS fabs
#include "oracle_impl.h"
