from .multiview_trainer import MultiviewTrainStep, FlatParams, shard_rays
from .validation import render, evaluate_psnr, save_pipeline, load_pipeline
