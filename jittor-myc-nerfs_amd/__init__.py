"""MI355X-native TensoRF volume renderer (package dir: jittor-myc-nerfs_amd; import as jittor_myc_nerfs_amd).

Drop-in surface of the reference's render path (tensorf-myc): TensorVMSplit / AlphaGridMask / MLPRender_Fea
(field.py), REFTensoRF / NerfPlusPlus (variants.py), OctreeRender_trilinear_fast (render.py), Blender-format ray generation (rays.py), backed by
hand-written HIP kernels for gfx950 behind the C-ABI in include/tvr.h (csrc/, built into lib/libtvr.so)."""
from .field import AlphaGridMask, MLPRender_Fea, TensorBase, TensorVMSplit, load_checkpoint  # noqa: F401
from .variants import Embedder, MLPNet, MLPRender_Fea_Ref, NerfPlusPlus, REFTensoRF  # noqa: F401
from .render import OctreeRender_trilinear_fast, N_to_reso, cal_n_samples, render_sharded, ShardedFramePipeline, FrameStream, shard_indices, shard_capacity, shard_gather_index, shard_send_views, shard_unpermute  # noqa: F401
from .evaluation import BlenderRays, evaluation, evaluation_path, rgb_ssim, rgb_ssim_torch  # noqa: F401
from .losses import TVLoss  # noqa: F401
from .training import GradBucket, make_graphed_step, shard_batch  # noqa: F401
from . import ngp, rays, synthetic  # noqa: F401
