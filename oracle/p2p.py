"""Prompt-to-Prompt attention control (oracle; see oracle/__init__.py).

A functional restatement of the reference's plug-in (text-guided/p2p/*):
  word_inds                 ptp_utils.py:297-315
  cross_alpha_table         ptp_utils.py:318-349
  replacement_mapper        seq_aligner.py:152-200
  refinement_mapper         seq_aligner.py:66-133
  equalizer                 ptp_controller_utils.py:92-104
  LocalBlend                ptp_classes.py:17-72
  Controller                ptp_classes.py:74-283  (AttentionControl/Store/ControlEdit/Replace/
                                                    Refine/Reweight folded into one class)
  make_controller           ptp_controller_utils.py:106-134
  P2PProcessor              ptp_utils.py:31-122
  register                  ptp_utils.py:277-295
"""
import numpy as np
import torch
import torch.nn.functional as F

MAXW = 77


# --------------------------------------------------------------------------- host-side tables
def word_inds(text, place, tok):
    """Token positions (1-based, BOS at 0) of a word (given by string or word index)."""
    words = text.split(" ")
    if isinstance(place, str):
        place = [i for i, w in enumerate(words) if w == place]
    elif isinstance(place, int):
        place = [place]
    out = []
    if len(place) > 0:
        pieces = [tok.decode([i]).strip("#") for i in tok.encode(text)][1:-1]
        acc, ptr = 0, 0
        for i, p in enumerate(pieces):
            acc += len(p)
            if ptr in place:
                out.append(i + 1)
            if acc >= len(words[ptr]):
                ptr += 1
                acc = 0
    return np.array(out)


def cross_alpha_table(prompts, num_steps, cross_steps, tok):
    """(num_steps+1, n_edit, 1, 1, 77) 0/1 table: which steps inject the source cross maps."""
    if not isinstance(cross_steps, dict):
        cross_steps = {"default_": cross_steps}
    if "default_" not in cross_steps:
        cross_steps["default_"] = (0.0, 1.0)
    n_edit = len(prompts) - 1
    tab = torch.zeros(num_steps + 1, n_edit, MAXW)

    def put(bounds, pi, inds=None):
        if isinstance(bounds, float):
            bounds = (0, bounds)
        s, e = int(bounds[0] * tab.shape[0]), int(bounds[1] * tab.shape[0])
        if inds is None:
            inds = torch.arange(MAXW)
        tab[:s, pi, inds] = 0
        tab[s:e, pi, inds] = 1
        tab[e:, pi, inds] = 0

    for i in range(n_edit):
        put(cross_steps["default_"], i)
    for key, item in cross_steps.items():
        if key == "default_":
            continue
        for i in range(n_edit):
            ind = word_inds(prompts[i + 1], key, tok)
            if len(ind) > 0:
                put(item, i, ind)
    return tab.reshape(num_steps + 1, n_edit, 1, 1, MAXW)


def _replacement_mapper_pair(x, y, tok):
    wx, wy = x.split(" "), y.split(" ")
    if len(wx) != len(wy):
        raise ValueError("attention replacement edit can only be applied on prompts with the "
                         f"same length but prompt A has {len(wx)} words and prompt B has {len(wy)} words.")
    diff = [i for i in range(len(wy)) if wy[i] != wx[i]]
    src = [word_inds(x, i, tok) for i in diff]
    tar = [word_inds(y, i, tok) for i in diff]
    m = np.zeros((MAXW, MAXW))
    i = j = 0
    cur = 0
    while i < MAXW and j < MAXW:
        if cur < len(src) and src[cur][0] == i:
            s_, t_ = src[cur], tar[cur]
            if len(s_) == len(t_):
                m[s_, t_] = 1
            else:
                for it in t_:
                    m[s_, it] = 1 / len(t_)
            cur += 1
            i += len(s_)
            j += len(t_)
        elif cur < len(src):
            m[i, j] = 1
            i += 1
            j += 1
        else:
            m[j, j] = 1
            i += 1
            j += 1
    return torch.from_numpy(m).float()


def replacement_mapper(prompts, tok):
    return torch.stack([_replacement_mapper_pair(prompts[0], p, tok) for p in prompts[1:]])


def _align(x, y):
    """Global alignment (gap 0, match +1, mismatch -1) with the reference's tie-breaking
    (seq_aligner.py:66-106): prefer 'left', then 'up', then 'diag'."""
    nx, ny = len(x), len(y)
    sc = np.zeros((nx + 1, ny + 1), dtype=np.int32)
    tb = np.zeros((nx + 1, ny + 1), dtype=np.int32)
    tb[0, 1:] = 1
    tb[1:, 0] = 2
    tb[0, 0] = 4
    for i in range(1, nx + 1):
        for j in range(1, ny + 1):
            left, up = sc[i, j - 1], sc[i - 1, j]
            diag = sc[i - 1, j - 1] + (1 if x[i - 1] == y[j - 1] else -1)
            best = max(left, up, diag)
            sc[i, j] = best
            tb[i, j] = 1 if best == left else (2 if best == up else 3)
    pairs = []
    i, j = nx, ny
    while i > 0 or j > 0:
        if tb[i, j] == 3:
            i, j = i - 1, j - 1
            pairs.append((j, i))
        elif tb[i, j] == 1:
            j -= 1
            pairs.append((j, -1))
        elif tb[i, j] == 2:
            i -= 1
        else:
            break
    pairs.reverse()
    return torch.tensor(pairs, dtype=torch.int64)


def _refinement_mapper_pair(x, y, tok):
    xs, ys = tok.encode(x), tok.encode(y)
    base = _align(xs, ys)
    alphas = torch.ones(MAXW)
    alphas[:base.shape[0]] = base[:, 1].ne(-1).float()
    mapper = torch.zeros(MAXW, dtype=torch.int64)
    mapper[:base.shape[0]] = base[:, 1]
    mapper[base.shape[0]:] = len(ys) + torch.arange(MAXW - len(ys))
    return mapper, alphas


def refinement_mapper(prompts, tok):
    ms, als = zip(*[_refinement_mapper_pair(prompts[0], p, tok) for p in prompts[1:]])
    return torch.stack(ms), torch.stack(als)


def equalizer(text, words, values, tok):
    if isinstance(words, (int, str)):
        words = (words,)
    eq = torch.ones(1, MAXW)
    for w, v in zip(words, values):
        eq[:, word_inds(text, w, tok)] = v
    return eq


# --------------------------------------------------------------------------- LocalBlend
def _word_layers(prompts, words, tok):
    al = torch.zeros(len(prompts), 1, 1, 1, 1, MAXW)
    for i, (p, ws) in enumerate(zip(prompts, words)):
        if isinstance(ws, str):
            ws = [ws]
        for w in ws:
            al[i, :, :, :, :, word_inds(p, w, tok)] = 1
    return al


class LocalBlend:
    """ptp_classes.py:17-72.  sub_words = the reference's substruct_words (:28-38): a second word mask, un-pooled and
    thresholded with th[1], whose complement multiplies the blend mask (:64-68)."""

    def __init__(self, prompts, num_steps, words, tok, start_blend=0.2, th=(0.3, 0.3), sub_words=None):
        self.alpha_layers = _word_layers(prompts, words, tok)
        self.sub_layers = None if sub_words is None else _word_layers(prompts, sub_words, tok)
        self.start_blend = int(start_blend * num_steps)
        self.counter = 0
        self.th = th

    def _mask(self, maps, layers, pool, x):
        m = (maps * layers).sum(-1).mean(1)                      # (2,1,16,16)
        if pool:
            m = F.max_pool2d(m, (3, 3), (1, 1), padding=(1, 1))
        m = F.interpolate(m, size=x.shape[2:])                   # nearest
        m = m / m.max(2, keepdim=True)[0].max(3, keepdim=True)[0]
        m = m.gt(self.th[0 if pool else 1])
        return m[:1] + m                                         # OR with the source mask

    def mask(self, maps, x):
        m = self._mask(maps, self.alpha_layers, True, x)
        if self.sub_layers is not None:
            m = m * ~self._mask(maps, self.sub_layers, False, x)
        return m.float()

    def __call__(self, x, store):
        self.counter += 1
        if self.counter > self.start_blend:
            maps = store["down_cross"][2:4] + store["up_cross"][:3]
            maps = torch.cat([t.reshape(self.alpha_layers.shape[0], -1, 1, 16, 16, MAXW)
                              for t in maps], dim=1)
            x = x[:1] + self.mask(maps, x) * (x - x[:1])
        return x


# --------------------------------------------------------------------------- controller
def _empty_store():
    return {f"{p}_{k}": [] for k in ("cross", "self") for p in ("down", "mid", "up")}


class Controller:
    """kind: 'store' (AttentionStore), 'replace', 'refine'; ``eq`` adds the Reweight wrapper."""

    def __init__(self, kind="store", prompts=None, num_steps=None, cross_steps=0.4,
                 self_steps=0.35, tok=None, local_blend=None, eq=None):
        self.kind = kind
        self.cur_step = 0
        self.cur_att_layer = 0
        self.num_att_layers = -1
        self.step_store = _empty_store()
        self.attention_store = {}
        self.local_blend = local_blend
        self.eq = eq
        if kind != "store":
            self.n_prompts = len(prompts)
            self.cross_alpha = cross_alpha_table(prompts, num_steps, cross_steps, tok)
            if isinstance(self_steps, float):
                self_steps = (0, self_steps)
            self.self_window = (int(num_steps * self_steps[0]), int(num_steps * self_steps[1]))
            if kind == "replace":
                self.mapper = replacement_mapper(prompts, tok)
            else:
                self.mapper, al = refinement_mapper(prompts, tok)
                self.alphas = al.reshape(al.shape[0], 1, 1, al.shape[1])

    # -- cross-map substitution rules (ptp_classes.py:241-243, 259-262, 279-283)
    def _replace_cross(self, base, repl):
        if self.kind == "replace":
            out = torch.einsum("hpw,bwn->bhpn", base, self.mapper)
        else:
            out = base[:, :, self.mapper].permute(2, 0, 1, 3) * self.alphas + repl * (1 - self.alphas)
        if self.eq is not None:
            out = out * self.eq[:, None, None, :]
        return out

    def _edit(self, cond, is_cross, place, save_attn):
        key = f"{place}_{'cross' if is_cross else 'self'}"
        if cond.shape[1] <= 32 ** 2 and save_attn:
            self.step_store[key].append(cond)                    # a VIEW: sees the edit below
        if self.kind == "store":
            return
        if is_cross or (self.self_window[0] <= self.cur_step < self.self_window[1]):
            h = cond.shape[0] // self.n_prompts
            a = cond.reshape(self.n_prompts, h, *cond.shape[1:])  # view of cond
            base, repl = a[0], a[1:]
            if is_cross:
                aw = self.cross_alpha[self.cur_step]
                a[1:] = self._replace_cross(base, repl) * aw + (1 - aw) * repl
            elif repl.shape[2] <= 32 ** 2:
                a[1:] = base.unsqueeze(0).expand(repl.shape[0], *base.shape)

    def __call__(self, probs, is_cross, place, save_attn):
        h = probs.shape[0]
        self._edit(probs[h // 2:], is_cross, place, save_attn)   # conditional half, in place
        if not save_attn:
            return probs
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            self._between_steps()
        return probs

    def _between_steps(self):
        if len(self.attention_store) == 0:
            self.attention_store = self.step_store
        else:
            for k in self.attention_store:
                for i in range(len(self.attention_store[k])):
                    self.attention_store[k][i] += self.step_store[k][i]
        self.step_store = _empty_store()

    def step_callback(self, x):
        if self.local_blend is not None:
            x = self.local_blend(x, self.attention_store)
        return x


def make_controller(prompts, is_replace, cross_steps, self_steps, blend_word=None, eq_params=None,
                    num_steps=None, tok=None):
    lb = None if blend_word is None else LocalBlend(prompts, num_steps, blend_word, tok)
    eq = None
    if eq_params is not None:
        eq = equalizer(prompts[1], eq_params["words"], eq_params["values"], tok)
    return Controller("replace" if is_replace else "refine", prompts, num_steps, cross_steps,
                      self_steps, tok, lb, eq)


# --------------------------------------------------------------------------- processor
class P2PProcessor:
    """Attention body with the controller hook between softmax and P.V (ptp_utils.py:65-122)."""

    def __init__(self, controller, place):
        self.controller = controller
        self.place = place

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None,
                 temb=None, use_controller=True, save_attn=True):
        residual = hidden_states
        x = hidden_states
        if attn.spatial_norm is not None:
            x = attn.spatial_norm(x, temb)
        nd = x.ndim
        if nd == 4:
            b, c, hh, ww = x.shape
            x = x.view(b, c, hh * ww).transpose(1, 2)
        if attn.group_norm is not None:
            x = attn.group_norm(x.transpose(1, 2)).transpose(1, 2)
        q = attn.to_q(x)
        is_cross = encoder_hidden_states is not None
        ctx = x if not is_cross else encoder_hidden_states
        if is_cross and attn.norm_cross:
            ctx = attn.norm_encoder_hidden_states(ctx)
        k, v = attn.to_k(ctx), attn.to_v(ctx)
        q, k, v = attn.head_to_batch_dim(q), attn.head_to_batch_dim(k), attn.head_to_batch_dim(v)
        probs = attn.get_attention_scores(q, k, attention_mask)
        if use_controller:
            self.controller(probs, is_cross, self.place, save_attn)
        o = attn.batch_to_head_dim(torch.bmm(probs, v))
        o = attn.to_out[1](attn.to_out[0](o))
        if nd == 4:
            o = o.transpose(-1, -2).reshape(b, c, hh, ww)
        if attn.residual_connection:
            o = o + residual
        return o / attn.rescale_output_factor


def register(model, controller):
    procs = {}
    for name in model.unet.attn_processors.keys():
        place = None
        for pre, pl in (("mid_block", "mid"), ("up_blocks", "up"), ("down_blocks", "down")):
            if name.startswith(pre):
                place = pl
        if place is None:
            continue
        procs[name] = P2PProcessor(controller, place)
    model.unet.set_attn_processor(procs)
    controller.num_att_layers = len(procs)
