"""TEST / A-B REFERENCE, not product: the CLIP towers as plain ``transformers`` modules behind the pipeline's ``text_encoder.encode`` /
``image_encoder.encode`` interfaces.  The product attaches the native towers (``anyv2v_amd.encoders.attach_native_clip_encoders``);
these wrappers are what the native towers are compared against (tests/test_host_logic.py, tests/gpu_checks.py::check_clip) and
lived in the product package until round 3 (VERDICT r3 housekeeping: a second backend does not belong there)."""
import torch
from PIL import Image

from anyv2v_amd.encoders import CLIP_MEAN, CLIP_STD, _center_crop_wide, _pil_to_tensor


class HFTextEncoder:
    """``transformers.CLIPTextModel`` + tokenizer behind the ``text_encoder.encode`` interface -- the reference's
    ``encode_prompt`` (``pipeline_i2vgen_xl.py:224-409``): pad / truncate to ``model_max_length``, and with
    ``clip_skip`` take hidden state ``-(clip_skip + 1)`` followed by the model's ``final_layer_norm`` (``:312-324``).
    These once-per-clip towers run as plain PyTorch-ROCm modules (they are not part of the HIP hot path)."""

    def __init__(self, model, tokenizer):
        self.model, self.tokenizer = model.eval(), tokenizer

    def to(self, device):
        self.model.to(device)
        return self

    @torch.no_grad()
    def encode(self, prompts, device, clip_skip=None):
        if isinstance(prompts, str):
            prompts = [prompts]
        ids = self.tokenizer(prompts, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                             return_tensors="pt").input_ids.to(device)
        self.model.to(device)
        if clip_skip is None:
            e = self.model(ids)[0]
        else:
            hs = self.model(ids, output_hidden_states=True).hidden_states
            ln = getattr(self.model, "text_model", self.model).final_layer_norm  # transformers 4.x nests it in .text_model
            e = ln(hs[-(clip_skip + 1)])
        return e.to(torch.float16)


class HFImageEncoder:
    """``transformers.CLIPVisionModelWithProjection`` behind ``image_encoder.encode`` -- ``_encode_image``
    (``pipeline_i2vgen_xl.py:411-441``) on the centre-cropped (w x w), bilinearly 224-resized frame (``:789-792``),
    normalised with the CLIP statistics (no rescale / crop / resize in the feature extractor)."""

    def __init__(self, model, crop=224):
        self.model, self.crop = model.eval(), crop

    def to(self, device):
        self.model.to(device)
        return self

    @torch.no_grad()
    def encode(self, image, width, device):
        img = _center_crop_wide(image, (width, width)).resize((self.crop, self.crop), resample=Image.BILINEAR)
        x = (_pil_to_tensor(img) + 1.0) * 0.5  # [1,3,224,224] in [0,1]
        mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
        std = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
        x = ((x - mean) / std).to(device=device, dtype=next(self.model.parameters()).dtype)
        self.model.to(device)
        return self.model(pixel_values=x).image_embeds[:, None].to(torch.float16)  # [1,1,1024]


def attach_hf_clip_encoders(pipe, root: str):
    """Load ``<root>/text_encoder``, ``<root>/tokenizer``, ``<root>/image_encoder`` (the sub-folders of the
    ``ali-vilab/i2vgen-xl`` checkpoint) with ``transformers`` when they exist locally.  Returns True when attached."""
    import os
    need = [os.path.join(root, d) for d in ("text_encoder", "tokenizer", "image_encoder")]
    if not all(os.path.isdir(d) for d in need):
        return False
    from transformers import CLIPTextModel, CLIPTokenizer, CLIPVisionModelWithProjection
    pipe.tokenizer = CLIPTokenizer.from_pretrained(need[1])
    pipe.text_encoder = HFTextEncoder(CLIPTextModel.from_pretrained(need[0], torch_dtype=torch.float16), pipe.tokenizer)
    pipe.image_encoder = HFImageEncoder(CLIPVisionModelWithProjection.from_pretrained(need[2], torch_dtype=torch.float16))
    pipe.feature_extractor = object()
    return True
