"""Device-side input preparation: what the reference's dataset does on the CPU between the
decoded uint8 image and the two tensors the model consumes.

Mirrors ``read_image`` (estimator/datasets/general_dataset.py:22-47) *after the decode* and the tensor part of
``ImageDataset.__getitem__`` (:188-219): ``img / 255.0`` (float64) -> bicubic ``align_corners=True`` resize to
``image_resolution`` -> ``to_tensor(...).float()`` = ``image_hr`` [3,H,W]; ``image_lr`` = bilinear
``align_corners=True`` resize of ``image_hr`` to the network input (depth_anything/transform.py:127-129).
The uint8 image is uploaded once (3 bytes / pixel instead of the 12 bytes / pixel float image the reference moves
with ``.cuda()``), both resizes run as HIP kernels.  There is no CPU path.
"""
import numpy as np
import torch


class ImagePreprocessor:
    def __init__(self, image_resolution=(2160, 3840), process_shape=(392, 518), dataset_name="general", device="cuda", ops=None):
        if ops is None:
            from .hip_ops import ops as _ops        # fails loudly when the HIP extension is missing
            ops = _ops
        self.ops = ops
        self.image_resolution = tuple(int(v) for v in image_resolution)
        self.process_shape = tuple(int(v) for v in process_shape)
        self.dataset_name = dataset_name
        self.device = torch.device(device)

    def __call__(self, img_u8):
        """img_u8: decoded image, uint8 [H,W,3] (numpy or torch; RGB - or the raw BGR file order for dataset 'u4k',
        general_dataset.py:24-25) -> dict(image_hr [3,H',W'] f32, image_lr [3,h,w] f32) on the device."""
        if isinstance(img_u8, np.ndarray):
            img_u8 = torch.from_numpy(np.ascontiguousarray(img_u8))
        if img_u8.dtype != torch.uint8 or img_u8.dim() != 3 or img_u8.shape[2] != 3:
            raise ValueError(f"expected a uint8 [H,W,3] image, got {img_u8.dtype} {tuple(img_u8.shape)}")
        img_u8 = img_u8.to(self.device, non_blocking=True).contiguous()
        u4k = self.dataset_name == "u4k"
        H, W = (img_u8.shape[0], img_u8.shape[1]) if u4k else self.image_resolution   # 'u4k' raw files are never resized
        image_hr = torch.empty((3, H, W), dtype=torch.float32, device=self.device)
        self.ops.u8_bicubic_to_f32(img_u8, image_hr, reverse_channels=u4k)
        image_lr = torch.empty((3,) + self.process_shape, dtype=torch.float32, device=self.device)
        for c in range(3):
            self.ops.resize_bilinear_f32(image_hr[c], image_lr[c])
        return {"image_hr": image_hr, "image_lr": image_lr}
