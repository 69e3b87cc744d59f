from .base_tracer import BaseTracer
from .packed_rf_tracer import PackedRFTracer
from .packed_sdf_tracer import PackedSDFTracer
