from .multiview_trainer import MultiviewTrainStep, FlatParams, shard_rays
from .sdf_trainer import SDFTrainStep
from .validation import render, evaluate_psnr, save_pipeline, load_pipeline
