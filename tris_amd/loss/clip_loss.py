"""`clip_forward` with the reference signature (loss/clip_loss.py:5-20 = train_stage1.py:263-278):
cosine similarity between aux-CLIP image features of `images` and text features of `tokenized_text` -> [N,1,1]."""
from .. import ops


def clip_forward(clip_model, images, tokenized_text):
    f_i = ops.l2norm(clip_model.encode_image(images))
    f_t = ops.l2norm(clip_model.encode_text(tokenized_text)[1])
    N, C = f_i.shape
    return ops.bmm(f_i.reshape(N, 1, C), f_t.reshape(N, 1, C), tB=True)
