from .raygen import (generate_default_grid, generate_centered_pixel_coords, generate_pinhole_rays, generate_ortho_rays,
                     LookAtCamera)
