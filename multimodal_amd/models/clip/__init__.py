from .image_encoder import CLIPViTEncoder  # noqa: F401
from .model import CLIP, CLIPOutput, clip_vit_b16, clip_vit_b32, clip_vit_l14  # noqa: F401
from .text_encoder import CLIPTextEncoder  # noqa: F401
