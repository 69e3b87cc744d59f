from .multiview_trainer import MultiviewTrainStep, FlatParams, shard_rays
