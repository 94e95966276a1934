"""TEST INFRASTRUCTURE -- not product code.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may
import this package; the product path (h-edit_amd/) never does and fails loudly without its HIP library.

fp32 torch restatements of the three reward networks of the guidance closures, evaluated on the PARAMETER CONTAINERS
of the product classes (hedit.clip_guidance.CLIPEncoder, hedit.arcface.IDLoss, hedit.arcface.lpips_loss.LPIPS_Loss:
same state_dict names as the reference's checkpoints), so that the native executors (csrc/vit.hip, irse.hip,
lpips.hip) have a CPU reference with the very same weights:

  * CLIP ViT prefix + Gram residual: text-guided-n-style/clip_guidance/clip/model.py:153-190 (LayerNorm in fp32,
    QuickGELU, ResidualAttentionBlock), :339-365 (encode_image_with_features up to block 3),
    clip_guidance/base_clip.py:38-66 (normalisation, get_gram_matrix_residual).
    PINNED on tests/golden/g10_clip.npz (the reference's CLIPEncoder run in the build container).
  * ArcFace IR-SE50 + IDLoss: face-swapping/arcface/facial_recognition/model_irse.py:9-48, helpers.py:28-119
    (bottleneck_IR_SE, SEModule), arcface/arcface_model.py:40-67 (crop, pool, cosine).
    PINNED on tests/golden/g12_idloss.npz (the reference's IDLoss run in the build container).
  * LPIPS-VGG16: lpips==0.1.4 (third party, absent from the reference tree and from this image) as
    arcface_model.py:69-94 calls it: ScalingLayer -> VGG16 taps relu1_2 ... relu5_3 -> channel unit-normalisation
    (eps 1e-10) -> squared difference -> bias-free 1x1 lin layers -> spatial mean -> sum.  PARITY UNPINNED: no
    reference vector exists for this network."""
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------- CLIP ViT prefix
def _ln(ln, x):                      # LayerNorm computed in fp32 whatever the stream dtype (model.py:153-159)
    return F.layer_norm(x.float(), ln.normalized_shape, ln.weight.float(), ln.bias.float(), ln.eps).to(x.dtype)


def _vit_block(blk, x):              # ResidualAttentionBlock (model.py:167-190), x: (N, L, D)
    N, L, D = x.shape
    h = blk.heads
    qkv = F.linear(_ln(blk.ln_1, x), blk.attn.in_proj_weight, blk.attn.in_proj_bias)
    q, k, v = (t.reshape(N, L, h, D // h).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
    p = torch.softmax((q * (D // h) ** -0.5) @ k.transpose(-1, -2), dim=-1)
    a = (p @ v).transpose(1, 2).reshape(N, L, D)
    x = x + F.linear(a, blk.attn.out_proj.weight, blk.attn.out_proj.bias)
    y = F.linear(_ln(blk.ln_2, x), blk.mlp.c_fc.weight, blk.mlp.c_fc.bias)
    return x + F.linear(y * torch.sigmoid(1.702 * y), blk.mlp.c_proj.weight, blk.mlp.c_proj.bias)


def vit_block_features(prefix, x):
    """ClipVisualPrefix container, x (N,3,S,S) CLIP-normalised -> token features after the last kept block, (N, L, D)
    (the reference's ``feats[layers-1]`` is the same tensor in (L, N, D) layout).  conv1 has kernel = stride = patch:
    one linear map per non-overlapping patch, written as unfold + matmul."""
    v = prefix.visual
    x = x.type(prefix.dtype)
    N, C, H, W = x.shape
    p = v.conv1.kernel_size[0]
    x = x.reshape(N, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(N, (H // p) * (W // p), C * p * p)
    x = x @ v.conv1.weight.reshape(v.conv1.weight.shape[0], -1).t()
    cls = v.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype, device=x.device)
    x = torch.cat([cls, x], dim=1) + v.positional_embedding.to(x.dtype)
    x = _ln(v.ln_pre, x)
    for blk in v.transformer.resblocks:
        x = _vit_block(blk, x)
    return x


def _clip_tokens(enc, im):           # batch item 0, class token dropped; Gram matrices accumulated in fp32
    return vit_block_features(enc.clip_model, im)[0, 1:, :].float()


def clip_gram_ref(enc):
    with torch.no_grad():
        f = _clip_tokens(enc, enc.ref)
        return torch.mm(f.t(), f)


def clip_gram_residual(enc, im1):
    """CLIPEncoder.get_gram_matrix_residual (base_clip.py:55-66): Gram matrix (D x D) of the block-3 patch tokens of
    ``im1`` (in [-1, 1], any size) minus that of the style reference; differentiable w.r.t. ``im1``."""
    im1 = F.interpolate(im1, size=(enc.size, enc.size), mode="bicubic")
    f = _clip_tokens(enc, enc.preprocess(im1))
    return torch.mm(f.t(), f) - clip_gram_ref(enc)


def clip_gram_residuals(enc, ims):
    """batched form: (N,3,H,W) -> (N,D,D), row i == clip_gram_residual(enc, ims[i:i+1])"""
    ims = F.interpolate(ims, size=(enc.size, enc.size), mode="bicubic")
    f = vit_block_features(enc.clip_model, enc.preprocess(ims))[:, 1:, :].float()
    return f.transpose(1, 2) @ f - clip_gram_ref(enc)


def clip_gram_residual_norms(enc, ims):
    return torch.linalg.norm(clip_gram_residuals(enc, ims), dim=(1, 2))


# ----------------------------------------------------------------------------------------- ArcFace IR-SE50
def _bn(m, x):
    return F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, False, 0.0, m.eps)


def _conv(m, x):
    return F.conv2d(x, m.weight, m.bias, m.stride, m.padding)


def _se(m, x):                       # SEModule (helpers.py:28-46)
    s = torch.sigmoid(_conv(m.fc2, F.relu(_conv(m.fc1, x.mean((2, 3), keepdim=True)))))
    return x * s


def _ir_unit(u, x):                  # bottleneck_IR_SE (helpers.py:97-119)
    if u.shortcut_layer is not None:
        sc = _bn(u.shortcut_layer[1], _conv(u.shortcut_layer[0], x))
    else:
        sc = x[:, :, ::u.stride, ::u.stride]          # MaxPool2d(1, stride)
    r = u.res_layer
    y = _conv(r[1], _bn(r[0], x))
    y = F.prelu(y, r[2].weight)
    y = _bn(r[4], _conv(r[3], y))
    return _se(r[5], y) + sc


def irse50_features(facenet, x):
    """Backbone container (model_irse.py:9-48), x (B,3,112,112) -> l2-normalised (B,512)"""
    il = facenet.input_layer
    x = F.prelu(_bn(il[1], _conv(il[0], x)), il[2].weight)
    for u in facenet.body:
        x = _ir_unit(u, x)
    ol = facenet.output_layer
    x = _bn(ol[0], x).flatten(1)                      # (Dropout is the identity in eval mode)
    x = F.linear(x, ol[3].weight, ol[3].bias)
    m = ol[4]
    x = F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, False, 0.0, m.eps)
    return x / torch.norm(x, 2, 1, True)


def idloss_extract_feats(idl, x):
    """IDLoss.extract_feats (arcface_model.py:40-47): pool to 256 x 256 if needed, crop the face region, pool to 112"""
    if x.shape[2] != 256:
        x = F.adaptive_avg_pool2d(x, (256, 256))
    x = x[:, :, 35:223, 32:220]
    return irse50_features(idl.facenet, F.adaptive_avg_pool2d(x.float(), (112, 112)))


def idloss_cosine_sim(idl, image):
    with torch.no_grad():
        ref = F.normalize(idloss_extract_feats(idl, idl.ref.to(image.device)), p=2, dim=-1)
    img = F.normalize(idloss_extract_feats(idl, image), p=2, dim=-1)
    return F.cosine_similarity(ref, img, dim=-1)


def idloss_cosine_loss(idl, image):
    return (1 - idloss_cosine_sim(idl, image)).mean()


# ----------------------------------------------------------------------------------------- LPIPS-VGG16 (parity unpinned)
_SLICE = (1, 1, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5)
_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)          # torchvision vgg16.features indices of the convolutions
_TAPS = {1: 0, 3: 1, 6: 2, 9: 3, 12: 4}                           # conv index -> lin index (a 2x2 max pool follows taps 0..3)


def lpips_features(net, x):
    h = (x - net.scaling_layer.shift) / net.scaling_layer.scale
    outs = []
    for l, (sl, idx) in enumerate(zip(_SLICE, _IDX)):
        c = getattr(getattr(net.net, f"slice{sl}"), str(idx))
        h = F.relu(F.conv2d(h, c.weight, c.bias, padding=1))
        if l in _TAPS:
            outs.append(h)
            if _TAPS[l] < 4:
                h = F.max_pool2d(h, 2, 2)
    return outs


def lpips_distance(net, x, y):
    """LPIPSNet container -> (B,1,1,1) distances, lpips.LPIPS(net='vgg').forward(x, y) in eval mode"""
    val = 0
    for t, (fx, fy) in enumerate(zip(lpips_features(net, x), lpips_features(net, y))):
        nx = fx / (fx.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        ny = fy / (fy.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        w = getattr(net, f"lin{t}").model[1].weight
        val = val + F.conv2d((nx - ny) ** 2, w).mean((2, 3), keepdim=True)
    return val


def lpips_loss(lp, x):
    """LPIPS_Loss.get_lpips_loss (arcface_model.py:88-94)"""
    return lpips_distance(lp.lpips_loss, x.float(), lp.src.to(x.device)).mean()
