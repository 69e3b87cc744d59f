from .multiview_trainer import MultiviewTrainStep, FlatParams
