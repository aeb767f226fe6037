"""The training hot loop: restatement of ``run_one_epoch`` / ``calculate_set_accuracy`` (main_both.py:81-198) over the
HIP module, plus the fused single-step form the benchmark times.

Quirks of the reference that change results and are reproduced on purpose:
  (i)  ``loss.backward()`` runs BEFORE ``loss = loss / acc_steps`` (main_both.py:112-114): gradients of the
       ``acc_steps`` micro-batches are SUMMED, only the logged loss is scaled;
  (ii) the optimizer steps every ``acc_steps`` batches or on the last batch; ``acc_steps == 0`` steps every batch;
  (iii) the per-batch loss is brought to the host every batch (``loss.cpu()``, :128).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch

import os

from . import lib as L
from .engine import make_text_pack

# run the text encoder on the live tokens only (engine.TextPack); "0" keeps the padded [B*T] layout
PACK_TEXT = os.environ.get("MMRCA_PACK_TEXT", "1") == "1"

mode_config_dict = {      # main_both.py:43-47
    'image_only': {"remove_text": True, "remove_image": False},
    'text_only': {"remove_text": False, "remove_image": True},
    'both': {"remove_text": False, "remove_image": False},
}


def get_class_weights_from_counts(counts: Sequence[int]) -> List[float]:
    """w_c = N_total / (n_classes * n_c)   (main_both.py:61-78)."""
    total = float(sum(counts))
    return [total / (len(counts) * c) for c in counts]


class FusedCrossEntropy:
    """torch.nn.CrossEntropyLoss(weight, label_smoothing) (main_both.py:87-93) as one HIP kernel that also emits
    d loss / d logits, so the step needs no autograd graph."""

    def __init__(self, weight: Optional[torch.Tensor] = None, label_smoothing: float = 0.0):
        self.weight, self.label_smoothing = weight, float(label_smoothing)

    def __call__(self, logits: torch.Tensor, labels: torch.Tensor):
        B, C = logits.shape
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dlogits = torch.empty_like(logits)
        L.xent_fwd_bwd(logits, labels.to(torch.int32), self.weight, self.label_smoothing, loss, dlogits, B, C, 1.0)
        return loss, dlogits


def _enqueue_step(model, ids, mask, images, labels, criterion, grad_sync, do_step, text_pack, seed, tt, ti):
    """forward -> loss -> backward (the gradient exchange rides on the backward) of one prepared batch on the current stream; no host
    sync, no host-side randomness -- which is what makes the sequence capturable (GraphedTrainStep).  Returns the device loss."""
    eng = model.engine
    logits = eng.forward(ids, mask, images, model.drop_ratio if model.training else 0.0, seed, save=(tt or ti),
                         enc_drop_p=(model.enc_dropout if model.training else 0.0), text_pack=text_pack, bn_train=model.training)
    loss, dlogits = criterion(logits, labels)
    if grad_sync is not None and grad_sync.world > 1 and criterion.weight is not None:
        # class-weighted loss under data parallelism: every rank normalised by ITS sum of w[y]; the all-reduce AVERAGES the
        # ranks, so rescale to the weighted mean over the GLOBAL batch (what the reference's single-process DataParallel step
        # computes): g_r * S_r * world / sum_r S_r.  Device-side scalar, no host sync.
        import torch.distributed as dist
        s_r = criterion.weight[labels.long()].sum()
        s_tot = s_r.clone()
        dist.all_reduce(s_tot)
        dlogits = dlogits * (s_r * grad_sync.world / s_tot)
    if grad_sync is not None:
        grad_sync.enabled = bool(do_step)
    eng.grad_sync = grad_sync
    eng.backward(dlogits, train_text=tt, train_image=ti)
    return loss


def _prepare_step(model, ids, mask, images, text_pack):
    """the host-side part of a step: modality dropout (the reference's numpy draws, multimodal_model.py:420-455), the trainable
    flags and the step's mask seed"""
    model._images, model._input_ids, model._attention_mask = images, ids, mask
    model.drop_modalities(False, False, False)
    if model._input_ids is not ids:
        text_pack = None                       # modality dropout zeroed the captions: the pack no longer describes them
    tt, ti = model._train_flags()
    model._fwd_count += 1
    return model._input_ids, model._attention_mask, model._images, text_pack, model._drop_seed + model._fwd_count, tt, ti


def hip_train_step(model, ids, mask, images, labels, criterion: FusedCrossEntropy, optimizer=None, grad_sync=None,
                   do_step: bool = True, text_pack=None):
    """forward -> loss -> backward (-> gradient all-reduce) -> optimizer step -> zero grads, all on the current
    stream without a host sync.  Returns the device loss tensor.
    text_pack: ``engine.make_text_pack(host_mask, device)`` of this batch -- the text encoder then skips the padding rows."""
    ids, mask, images, text_pack, seed, tt, ti = _prepare_step(model, ids, mask, images, text_pack)
    loss = _enqueue_step(model, ids, mask, images, labels, criterion, grad_sync, do_step, text_pack, seed, tt, ti)
    if do_step and optimizer is not None:
        if grad_sync is not None:
            grad_sync.finish()
        optimizer.step()
        optimizer.zero_grad()
    return loss


class GraphedTrainStep:
    """``hip_train_step`` with the forward -> loss -> backward launches of a batch shape captured ONCE in a HIP graph and replayed
    (the launch-bound regime).  A small-batch step is ~1,000 launches of a few microseconds each -- configs[0]
    (ShuffleNetV2 + DistilBERT, B = 4): 14 ms of Python + ctypes + hipLaunchKernel per step against 11 ms of kernels -- and the
    reference pays the same from its own Python loop (main_both.py:81-134); one hipGraphLaunch replaces them.

    What a replay must not freeze, and how it does not:
      * dropout masks -- seeds are launch arguments; the graph's first node loads the replay's distance to the captured step into
        the kernels' mask epoch (lib.seed_epoch_set), its last node resets it: replay r draws the masks of eager step s + r;
      * inputs -- copied into the static buffers the captured launches read;
      * modality dropout and the trainable flags -- decided on the host BEFORE the graph (a different flag set is a different graph);
      * the optimizer -- stays outside (2-4 launches): learning-rate schedules and AdamW's step count are host values;
      * stochastic depth (EfficientNetV2) -- a counter-based draw of the step seed like the dropout masks (mmrca_sd_rowscale).
    Captions run in the padded layout (a packed layout changes launch shapes per batch).  The first `warmup` calls of a shape run
    eagerly (they allocate the engine's buffers), the next one captures.
    Data parallel: on RCCL the gradient exchange is captured WITH the step -- the bucketed all-reduces the backward launches span by
    span (and the wait in front of the optimizer) become graph nodes on RCCL's stream, so an 8-rank node replays eight graphs instead of
    running eight ~20-ms Python enqueue loops per step; stepping and accumulating micro-batches are different graphs (only the former
    exchanges).  On gloo (CPU tests), or with MMRCA_GRAPH_DP=0, a multi-rank step stays eager."""

    def __init__(self, model, criterion: FusedCrossEntropy, optimizer=None, grad_sync=None, warmup: int = 2):
        self.model, self.criterion, self.optimizer, self.grad_sync, self.warmup = model, criterion, optimizer, grad_sync, int(warmup)
        self._graphs, self._eager_left, self._no_graph = {}, {}, set()
        self._epoch = torch.zeros(1, dtype=torch.int64, device=model.engine.device)
        self.replays = 0
        self._generation = getattr(model.engine, "buffer_generation", 0)
        L.seed_epoch_set(0)                    # resolves the epoch words' addresses outside any capture

    def _sync_in_graph(self):
        gs = self.grad_sync
        return gs is not None and gs.active() and gs.capturable()

    def _eager(self):
        gs = self.grad_sync
        return gs is not None and gs.active() and not gs.capturable()

    def __call__(self, ids, mask, images, labels, do_step: bool = True):
        model, eng = self.model, self.model.engine
        ids, mask, images, _, seed, tt, ti = _prepare_step(model, ids, mask, images, None)
        in_graph = self._sync_in_graph()
        key = (tuple(ids.shape), tuple(images.shape), images.dtype, labels.dtype, tt, ti, bool(model.training),
               float(model.drop_ratio), float(model.enc_dropout), bool(do_step) if in_graph else None)
        if getattr(eng, "buffer_generation", 0) != self._generation:       # engine.release_buffers() freed what the graphs point into
            self._graphs.clear()
            self._eager_left.clear()
            self._generation = getattr(eng, "buffer_generation", 0)
        ent = self._graphs.get(key)
        if ent is None and key not in self._no_graph and not self._eager() and self._eager_left.setdefault(key, self.warmup) <= 0:
            try:
                ent = self._capture(key, ids, mask, images, labels, seed, tt, ti, do_step)
                captured_now = True
            except RuntimeError as e:          # a step that cannot be captured keeps working, launched from Python
                # ... but a kernel / argument error of the library is an error, not a slowdown -- unless it is the capture itself
                # speaking through a launch check ("operation not permitted when stream is capturing", "... previous error during
                # capture"): that one is a capture failure like torch's own
                if isinstance(e, L.MmrcaError) and "captur" not in str(e).lower():
                    raise
                print(f"HIP graph capture failed for batch shape {key[0]} x {key[1]} ({type(e).__name__}: {str(e)[:200]}); this shape stays eager")
                self._no_graph.add(key)
                L.seed_epoch_set(0)
        else:
            captured_now = False
        if ent is None:
            if key in self._eager_left:
                self._eager_left[key] -= 1
            loss = _enqueue_step(model, ids, mask, images, labels, self.criterion, self.grad_sync, do_step, None, seed, tt, ti)
        else:
            if captured_now:
                pass                           # (_capture cloned this call's inputs into the static buffers)
            else:
                eng.refresh_working_copy()     # (a no-op unless parameters were loaded since the last step)
                for dst, src in zip(ent["inputs"], (ids, mask, images, labels)):
                    dst.copy_(src, non_blocking=True)
            self._epoch.fill_(seed - ent["seed"])
            ent["graph"].replay()
            self.replays += 1
            if eng.conv is not None and model.training and ent["replayed"]:
                eng.conv.n_train_forwards += 1          # (the capturing call ran conv.forward's host side once already)
            ent["replayed"] = True
            loss = ent["loss"].clone()         # the graph's loss word is rewritten by the next replay
        if do_step and self.optimizer is not None:
            if self.grad_sync is not None:
                self.grad_sync.finish()
            self.optimizer.step()
            self.optimizer.zero_grad()
        return loss

    def _capture(self, key, ids, mask, images, labels, seed, tt, ti, do_step=True):
        model, eng = self.model, self.model.engine
        eng.refresh_working_copy()
        inputs = [t.clone() for t in (ids, mask, images, labels)]
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # host-side state the captured (never executed) enqueue advances: restored if the capture fails, so that the eager step that
        # follows starts from where this call started
        conv = eng.conv
        snap = dict(saved=eng._saved, bufs=set(eng._bufs), n_fwd=(conv.n_train_forwards if conv is not None else 0),
                    conv_saved=(conv.saved if conv is not None else None), conv_bufs=(set(conv._bufs) if conv is not None else set()),
                    no_arena=(conv is not None and conv._bn_arena is None))
        try:
            # thread_local: only THIS thread's HIP calls are illegal while capturing.  The DataLoader's pin-memory thread
            # (main_both.py's loaders: num_workers 16, pin_memory) calls hipHostMalloc / event queries meanwhile; in the default
            # "global" mode those would invalidate the capture or raise in that thread.  The engine launches from this thread only.
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                L.seed_epoch_set(device_value=self._epoch)
                gs = self.grad_sync if self._sync_in_graph() else None
                loss = _enqueue_step(model, inputs[0], inputs[1], inputs[2], inputs[3], self.criterion, gs, do_step if gs is not None else True,
                                     None, seed, tt, ti)
                if gs is not None and do_step:
                    gs.finish()                # RCCL's stream joins the capturing stream: the graph ends with averaged gradients
                L.seed_epoch_set(0)
        except BaseException:
            # buffers first created inside the dead capture were "zeroed" by a fill that never ran (padding rows are assumed zero by
            # the Mk-rounded weight-gradient GEMMs): drop them, the eager step allocates them afresh
            for k in set(eng._bufs) - snap["bufs"]:
                del eng._bufs[k]
            eng._saved = snap["saved"]
            if conv is not None:
                for k in set(conv._bufs) - snap["conv_bufs"]:
                    del conv._bufs[k]
                conv.n_train_forwards, conv.saved = snap["n_fwd"], snap["conv_saved"]
                if snap["no_arena"]:           # (the BatchNorm arena itself was born in the dead capture)
                    conv._bn_arena, conv._bn_off, conv._bn_used = None, {}, 0
            if self.grad_sync is not None:     # collectives recorded into the dead capture never ran
                self.grad_sync.pending.clear()
                self.grad_sync._acc_lo = self.grad_sync._acc_hi = None
            raise
        print(f"HIP graph captured for the train step of batch shape {key[0]} x {key[1]} (trainable text / image encoder: {tt} / {ti})")
        ent = dict(graph=g, inputs=inputs, loss=loss, seed=seed, replayed=False)
        self._graphs[key] = ent
        return ent


def trim_caption_columns(tokens, mask, multiple: int = 16, width=None):
    """Captions as the dataset pads them (CustomImageTextFolder.py:305-333: ``padding='max_length'`` with the text model's maximum, 512)
    cut down to the columns the batch uses, rounded up to `multiple`: key columns past every caption's end are masked for every query and
    the class-token pooling never reads their rows, so the logits and gradients are those of the padded batch.  The eager step gets the
    same effect from the packed layout (engine.TextPack); a HIP-graph replay needs static shapes, so it runs padded rows -- the
    reference's file-name captions are ~10 tokens: 512 padded columns would be 50x the text encoder's work -- and the trimmed width
    (16, 32, ...) is part of the graph key: a handful of graphs.  Host tensors in, host views out (no device sync).
    width: use this width instead of the batch's own (data parallel: the maximum over the ranks, distributed.agree_caption_width)."""
    T = int(mask.shape[1])
    t_eff = caption_width(mask, multiple) if width is None else min(T, int(width))
    if t_eff >= T:
        return tokens, mask
    return tokens[:, :t_eff].contiguous(), mask[:, :t_eff].contiguous()


def caption_width(mask, multiple: int = 16) -> int:
    """the columns a batch of padded captions uses, rounded up to `multiple` (never more than the padded width)"""
    T = int(mask.shape[1])
    used = (mask != 0).any(0).nonzero()
    last = int(used.max()) + 1 if used.numel() else 1
    return min(T, (last + multiple - 1) // multiple * multiple)


def stage_images(raw, hw_device, image_pipeline=None, aug_params=None):
    """Batch images as the DataLoader delivers them -> fp32 [B,3,H,W] in HBM.  A tensor is the per-sample CPU transform's
    output (copied over); decoded uint8 HWC images (a list, or ``collate_decoded``'s packed batch) go through the GPU
    pipeline (preprocess.GpuImagePipeline): pad / resize / (augment) / normalise as a handful of launches per batch."""
    if isinstance(raw, (list, tuple, dict)):
        n = len(raw["shapes"]) if isinstance(raw, dict) else len(raw)
        if image_pipeline is None:
            raise ValueError("decoded image lists need an image_pipeline (main_both.py --gpu_preprocess)")
        return image_pipeline(raw, aug=aug_params(n) if aug_params is not None else None)
    return raw.to(hw_device, non_blocking=True)


def run_one_epoch(epoch_num, model, data_loader, len_train_data, hw_device, batch_size, train_optimizer, weights,
                  use_class_weights, acc_steps, smoothing, grad_sync=None, verbose=True, image_pipeline=None, aug_params=None,
                  hip_graph=None):
    """main_both.py:81-134 (same argument order).  Works with any model exposing the MM_RCA forward; with the HIP
    module the criterion is the fused kernel and the backward is the engine's.
    hip_graph: a dict the caller keeps across epochs (--hip_graph): the steps then replay HIP graphs (GraphedTrainStep), one per batch
    shape; the dict owns them together with the criterion (its class-weight tensor is read by the captured launches)."""
    batch_loss = []
    n_batches = math.ceil(len_train_data / batch_size)
    fused = hasattr(model, "engine")
    if use_class_weights:
        opt_weights = torch.tensor(weights, dtype=torch.float32, device=hw_device)     # (.cuda() in the reference, :89)
    else:
        opt_weights = None
    if fused:
        criterion = FusedCrossEntropy(opt_weights, smoothing)
    else:
        criterion = torch.nn.CrossEntropyLoss(weight=opt_weights, label_smoothing=smoothing).to(hw_device)
    graphed = None
    if fused and hip_graph is not None:
        gkey = (tuple(float(w) for w in weights) if use_class_weights else None, float(smoothing), id(train_optimizer), id(grad_sync))
        graphed = hip_graph.get(gkey)
        if graphed is None:
            graphed = hip_graph[gkey] = GraphedTrainStep(model, criterion, train_optimizer, grad_sync)
    n_loader = len(data_loader)
    for batch_idx, (data, labels) in enumerate(data_loader):
        images = stage_images(data['image']['raw_image'], hw_device, image_pipeline, aug_params)
        texts = data['text']
        # the mask is still on the host here: build the packed token layout without a device sync
        pack = make_text_pack(texts['attention_mask'], hw_device) if (fused and graphed is None and PACK_TEXT and not texts['attention_mask'].is_cuda) else None
        tok, msk = texts['tokens'], texts['attention_mask']
        if graphed is not None and not msk.is_cuda:
            if grad_sync is not None and grad_sync.world > 1:
                # every rank runs this batch at the SAME caption width (the maximum of theirs): the width is in the graph key, and
                # ranks keyed differently would warm up / capture / replay at different steps (host-side gloo all-reduce, no device sync)
                from .distributed import agree_caption_width
                tok, msk = trim_caption_columns(tok, msk, width=agree_caption_width(caption_width(msk)))
            else:
                tok, msk = trim_caption_columns(tok, msk)
        ids = tok.to(hw_device, non_blocking=True)
        mask = msk.to(hw_device, non_blocking=True)
        labels = labels.to(hw_device, non_blocking=True)
        if acc_steps != 0:
            do_step = ((batch_idx + 1) % acc_steps == 0) or (batch_idx + 1 == n_loader)
        else:
            do_step = True
        if graphed is not None:
            loss = graphed(ids, mask, images, labels, do_step)[0]
        elif fused:
            loss = hip_train_step(model, ids, mask, images, labels, criterion, train_optimizer, grad_sync, do_step, text_pack=pack)[0]
        else:
            out = model(_input_ids=ids, _attention_mask=mask, _images=images)
            loss = criterion(out, labels)
            loss.backward()
            if do_step:
                train_optimizer.step()
                train_optimizer.zero_grad()
        if acc_steps != 0:
            loss = loss / acc_steps            # logged value only (:114)
        if verbose:
            if do_step and acc_steps != 0:
                print("Optimizer step on batch idx: {}".format(batch_idx))
            print("Batch {}/{} on epoch {}".format(batch_idx, n_batches, epoch_num))
        # the reference copies every batch's loss to the host here (:131), which makes the host wait for the step it has just
        # queued; the values are only read after the epoch, so they stay on the device until then and the host is free to
        # stage the next batch (decode hand-over, descriptors, H2D) while the GPU works
        batch_loss.append(loss.detach())
    return n_batches, [l.cpu() for l in batch_loss]


def calculate_set_accuracy(model, data_loader, len_data, device, batch_size, mode, eval_mode, verbose=True,
                           all_reduce=None, n_real=None, image_pipeline=None, aug_params=None):
    """main_both.py:141-198.  Returns (accuracy %, sklearn classification report dict).
    n_real: only the first n_real samples this rank draws are scored (ShardedSampler.num_real: the rest is the wrap-around
    padding that equalises the ranks and would otherwise be counted twice)."""
    n_batches = math.ceil(len_data / batch_size)
    all_labels, all_predictions = [], []
    seen = 0
    with torch.no_grad():
        for batch_idx, (data, labels) in enumerate(data_loader):
            images = stage_images(data['image']['raw_image'], device, image_pipeline, aug_params)
            texts = data['text']
            tok, msk = texts['tokens'], texts['attention_mask']
            if hasattr(model, "engine") and not msk.is_cuda:
                # the dataset pads every caption to the text model's maximum (512, CustomImageTextFolder.py:305-333) while the
                # captions are file-name stems of ~10 tokens: the columns no caption of the batch uses are dropped on the host --
                # same logits (masked keys, class-token pooling), a fraction of the text encoder's work in the FOUR evaluation
                # passes of every epoch (main_both.py:596-646)
                tok, msk = trim_caption_columns(tok, msk)
            ids, mask = tok.to(device, non_blocking=True), msk.to(device, non_blocking=True)
            labels = labels.to(device, non_blocking=True)
            outputs = model(_input_ids=ids, _attention_mask=mask, _images=images, eval=eval_mode,
                            remove_text=mode["remove_text"], remove_image=mode["remove_image"])
            pred = torch.max(outputs, 1)[1].view(-1)
            if n_real is not None:
                keep = max(0, min(len(labels), n_real - seen))
                seen += len(labels)
                pred, labels = pred[:keep], labels[:keep]
            if verbose:
                print("Batches {}/{} ".format(batch_idx, n_batches))
            # (the reference copies the predictions of every batch to the host, :181-189 -- a device sync per batch; they are only read
            # after the pass, so they stay in HBM until then)
            all_labels.append(labels)
            all_predictions.append(pred)
    lab = torch.cat([t.view(-1) for t in all_labels]).cpu() if all_labels else torch.zeros(0, dtype=torch.int64)
    prd = torch.cat([t.view(-1) for t in all_predictions]).cpu() if all_predictions else torch.zeros(0, dtype=torch.int64)
    correct = int((lab == prd).sum().item())
    labels_flat = [int(x) for x in lab]
    preds_flat = [int(x) for x in prd]
    count = len_data
    if all_reduce is not None:
        correct, count = all_reduce(correct, len(labels_flat), device)
    try:
        from sklearn.metrics import classification_report
        report = classification_report(labels_flat, preds_flat, labels=[0, 1, 2, 3],
                                       target_names=["black", "blue", "green", "ttr"], output_dict=True, zero_division=0)
    except Exception:
        report = {}
    acc = 100 * (correct / max(count, 1))
    if verbose:
        print("Set acc: ", acc)
    return acc, report
