"""The augmentation kernels of ONE 1 024-image cfg2 batch (10 crops), once: target of the PMC passes (scratch/r4/pmc_aug.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
plane = rs.rand(3, 256, 256).astype(np.float32)
specs = [CropSpec(crop_size=224, num_crops=1, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=1.0, flip_prob=0.5),
         CropSpec(crop_size=224, num_crops=1, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=0.1, solarize_prob=0.2, flip_prob=0.5),
         CropSpec(crop_size=96, num_crops=8, crop_min_scale=0.05, crop_max_scale=0.25, jitter_prob=0.8, blur_prob=0.5, flip_prob=0.5)]
pipe = DeviceMultiCropPipeline(specs, dev, seed=1)
for _ in range(2):
    out = pipe([plane] * 1024)
torch.cuda.synchronize()
