from .base_trainer import (BaseTrainer, ConfigBaseTrainer, ConfigAdam, ConfigAdamW, ConfigRMSprop, ConfigDataloader,
                           instantiate_optimizer)
from .multiview_trainer import MultiviewTrainStep, MultiviewTrainer, ConfigMultiviewTrainer, FlatParams, shard_rays
from .sdf_trainer import SDFTrainStep, SDFTrainer, ConfigSDFTrainer
from .validation import render, evaluate_psnr, save_pipeline, load_pipeline
