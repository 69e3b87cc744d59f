from .rays import Rays
from .render_buffer import RenderBuffer
from .wisp_module import WispModule
