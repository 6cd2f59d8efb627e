"""Drop-in classes for the reference's point-cloud projection path, backed by libm355.

Names, constructor arguments, forward signatures, buffers and return conventions follow the reference
(file:line relative to /root/reference/code); keyword-only extras select the documented non-literal
behaviours and default to the literal ones.
"""
import torch
import torch.nn as nn

from . import ops


class CameraUtilities(object):
    """camera/coordinate_system_transformation.py:16-39"""

    def transformation_3d_coord_to_camera_coord(self, point_cloud, rotation, field_of_view, camera_view_distance):
        return ops.camera_transform(point_cloud, rotation, field_of_view, camera_view_distance)


class TrilinearInterpolation(object):
    """utils/trilinear_interpolation.py:11-74 (`size` is honoured; the reference module hard-wires 64 through its caller)"""

    def __init__(self, epsilon=1e-6, size=64, *, fixed_weights=False):
        self.epsilon = epsilon
        self.size = size
        self.fixed_weights = fixed_weights

    def get_point_cloud_object_borders(self, point_cloud):
        """tri:17-26"""
        return ((point_cloud < 0.5 - self.epsilon) & (point_cloud > -0.5 + self.epsilon)).all(dim=-1).view(-1)

    def get_grid(self, point_cloud, voxel_size):
        """tri:28-35"""
        return (voxel_size - 1) * (point_cloud + 0.5)

    def trilinear_interpolation(self, point_cloud):
        """tri:62-74: [B,N,3] camera-frame points (z,y,x) -> [B,S,S,S] occupancy in [0,1]"""
        if self.epsilon != 1e-6:
            raise ValueError("the HIP kernel implements the reference's epsilon=1e-6 in-bounds test")
        return ops.trilinear(point_cloud, self.size, self.fixed_weights)


class VoxelsSmooth(object):
    """utils/smooth_voxels.py:10-84"""

    def separate_kernels(self, std_dev, kernel_size=21, *, true_gaussian=False):
        """sm:14-42: three views [1,1,1,1,K], [1,1,1,K,1], [1,1,K,1,1] of the normalised 1-D kernel
        exp(+x^2/2s^2) (sign as written, defect D4) -- host arithmetic identical to the reference's."""
        a, b = (-kernel_size // 2, kernel_size // 2)
        x = torch.arange(a + 1.0, b + 1.0)
        e = pow(-x, 2) / (2 * pow(std_dev, 2))
        k = torch.exp(-e if true_gaussian else e)
        k = k / k.sum()
        return [k.view(1, 1, 1, 1, -1), k.view(1, 1, 1, -1, 1), k.view(1, 1, -1, 1, 1)]

    def smooth(self, voxels, kernels, scale=None, *, chained=False):
        """sm:44-84.  Literal behaviour (default): every kernel convolves the ORIGINAL volume and overwrites the
        result, so only the last kernel of the list takes effect (defect D5); chained=True applies them in sequence."""
        if len(kernels) == 0:
            raise ValueError("smooth() needs at least one kernel (the reference raises AttributeError here, defect D2)")
        items = []
        for k in kernels:
            shp = list(k.shape)
            ax5 = max(range(len(shp)), key=lambda i: shp[i])          # np.argmax(kernel.shape)
            items.append((k.reshape(-1).to(device=voxels.device, dtype=torch.float32).contiguous(), ax5 - 2))
        if not chained:
            items = items[-1:]
        return ops.smooth(voxels, [t for t, _ in items], [a for _, a in items], scale)


class EffectiveLossFunction(nn.Module):
    """utils/effective_loss_function.py:10-81.

    forward(point_cloud[B,N,3], rotation[B,4], scale=None|[B,1]) -> silhouette [B,S,S].

    Deviations from the file as committed, all documented in SURVEY.md section 8a ledger:
      * D2 (shim S1): the smoothing kernels are built from the `sigma` buffer and `kernel_size`
        (the reference passes kernels=() and raises);
      * D6: `voxel_size` is honoured (the reference hard-wires 64); the default stays 64.
    literal=True (default) keeps D3 (weights), D4 (inverted Gaussian) and D5 (depth-only smoothing).
    """

    def __init__(self, voxel_size=64, kernel_size=21, smooth_sigma=3.0, *, fixed_weights=False, true_gaussian=False):
        super(EffectiveLossFunction, self).__init__()
        self.voxel_size = voxel_size
        self.kernel_size = kernel_size
        self.register_buffer("sigma", torch.tensor(smooth_sigma))
        self.fixed_weights = fixed_weights
        self.true_gaussian = true_gaussian

    def _flags(self):
        f = ops.TAPS_FROM_SIGMA
        if self.fixed_weights:
            f |= ops.FIXED_WEIGHTS
        if self.true_gaussian:
            f |= ops.TRUE_GAUSSIAN
        from . import conv
        if conv.is_deterministic() and self.kernel_size == 21:   # (the 21-tap kernels carry the order-independent splat)
            f |= ops.DET_SPLAT
        return f

    def termination_probs(self, voxels, epsilon=1e-5):
        """elf:18-56: occupancies [B,D,H,W] -> ray termination probabilities [B,D+1,H,W] (last = background)"""
        return ops.termination_probs(voxels, epsilon)

    def forward(self, point_cloud, rotation, scale=None):
        sigma = self.sigma
        if sigma.device != point_cloud.device or sigma.dtype != torch.float32:
            sigma = sigma.to(device=point_cloud.device, dtype=torch.float32)
        return ops.project_silhouette(point_cloud, rotation, scale, sigma, self.kernel_size, self.voxel_size,
                                      self._flags())


class SupervisedLoss(nn.Module):
    """models/supervised_part.py:68-72: {"full_loss": sum((proj - bilinear_half(mask))^2) / (2B)}"""

    def forward(self, projection, masks, **kwargs):
        total, _ = ops.silhouette_sse(projection, masks)
        return dict(full_loss=total.reshape(()) / (2 * projection.size(0)))


def quaternion_addition(q1, q2):
    """quaternions/operations.py:15-40"""
    return torch.stack([a + b for a, b in zip(torch.unbind(q1, dim=-1), torch.unbind(q2, dim=-1))], dim=-1)


def quaternion_subtraction(q1, q2):
    """quaternions/operations.py:42-66"""
    return torch.stack([a - b for a, b in zip(torch.unbind(q1, dim=-1), torch.unbind(q2, dim=-1))], dim=-1)


def quaternion_square(q):
    """quaternions/operations.py:99-118: (s^2 - |v|^2, 2 s v).  The reference evaluates the squares with math.pow, which
    turns them into Python floats and makes its torch.stack raise for every input; the formula it spells out is
    implemented on tensors here."""
    s, x, y, z = torch.unbind(q, dim=-1)
    return torch.stack([s * s - x * x - y * y - z * z, 2 * s * x, 2 * s * y, 2 * s * z], dim=-1)


def quaternion_multiplication(q1, q2):
    """quaternions/operations.py:68-97 (tiny [.,4] tensors on the loss side; not a hot-path kernel)"""
    a0, a1, a2, a3 = torch.unbind(q1, dim=-1)
    b0, b1, b2, b3 = torch.unbind(q2, dim=-1)
    return torch.stack([a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3,
                        a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
                        a0 * b2 + a2 * b0 + a3 * b1 - a1 * b3,
                        a0 * b3 + a3 * b0 + a1 * b2 - a2 * b1], dim=-1)


def quaternion_conjugate(q):
    """quaternions/operations.py:120-136"""
    return q * q.new_tensor([1.0, -1.0, -1.0, -1.0])


class QuaternionOperations(object):
    """quaternions/operations.py:11-136 (loss-side helpers on small [..., 4] tensors; plain tensor arithmetic)"""

    def quaternion_addition(self, q1, q2):
        return quaternion_addition(q1, q2)

    def quaternion_subtraction(self, q1, q2):
        return quaternion_subtraction(q1, q2)

    def quaternion_multiplication(self, q1, q2):
        return quaternion_multiplication(q1, q2)

    def quaternion_square(self, q):
        return quaternion_square(q)

    def quaternion_conjugate(self, q):
        return quaternion_conjugate(q)


class PointsQuaternionsConverter(object):
    """quaternions/points_quaternions.py:12-35"""

    @staticmethod
    def points_to_quaternions(xyz_triplet):
        if xyz_triplet.size(-1) != 3:
            raise ValueError("points_to_quaternions: the last dimension must be 3")
        return torch.nn.functional.pad(xyz_triplet, (1, 0, 0, 0))


class PointsQuaternionsRotator(object):
    """quaternions/points_quaternions.py:37-81: rotate_points(xyz [B,N,3], q [B,4], inverse_rotation_direction) -> [B,N,3]
    on the HIP kernel of the camera transform's rotation stage (bit-exact with the reference's two Hamilton products).
    The reference's `assert len(xyz_triplet) == 3` (defect D1: batch size must be 3) is not reproduced."""

    @staticmethod
    def rotate_points(xyz_triplet, q, inverse_rotation_direction):
        return ops.rotate_points(xyz_triplet, q, inverse_rotation_direction)


class UnsupervisedLoss(nn.Module):
    """models/unsupervised_part.py:90-143.  Defect D8 (`self.num_candidates` undefined) is restated with
    `number_of_pose_predictor_candidates`, as SURVEY.md prescribes."""

    def __init__(self, number_of_pose_predictor_candidates=4, student_weight=20.00):
        super().__init__()
        self.student_weight = student_weight
        self.number_of_pose_predictor_candidates = number_of_pose_predictor_candidates
        self.minimum_indexes = None

    def forward(self, predictions, masks, training):
        projection, *poses = predictions
        K = self.number_of_pose_predictor_candidates
        if not training:
            total, _ = ops.silhouette_sse(projection, masks)
            return dict(projection_loss=total.reshape(()) / projection.size(0))
        ensemble_poses, student_poses = poses
        # (data parallel: a rank's batch must hold whole candidate groups -- parallel.shard_by_image; a cloud-granular shard would
        # silently take the argmin over a mixed group, so the shapes are checked here)
        from .parallel import check_image_groups
        check_image_groups(projection.size(0), masks.size(0), K)
        # masks are repeated K times per element (unsup:113); the kernel indexes mask row b // K instead
        _, sse = ops.silhouette_sse(projection, masks, mask_repeat=K)
        projection_loss = sse.view(-1, K)
        minimum_indexes = projection_loss.argmin(dim=-1).detach()
        batch_indexes = torch.arange(minimum_indexes.size(0), device=minimum_indexes.device)
        minimum_projection_loss = projection_loss[batch_indexes, minimum_indexes].sum() / minimum_indexes.size(0)
        ensemble_poses = ensemble_poses.view(-1, K, 4)
        best_poses = ensemble_poses[batch_indexes, minimum_indexes, :].detach()
        poses_difference = torch.nn.functional.normalize(
            quaternion_multiplication(best_poses, quaternion_conjugate(student_poses)), dim=-1)
        angle_difference = poses_difference[:, 0]
        student_loss = (1 - angle_difference ** 2).sum() / minimum_indexes.size(0)
        self.minimum_indexes = minimum_indexes.detach()
        total_loss = minimum_projection_loss + self.student_weight * student_loss
        return dict(projection_loss=minimum_projection_loss, student_loss=student_loss, total_loss=total_loss)
