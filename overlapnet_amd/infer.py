"""`Infer` -- drop-in for the reference's `src/two_heads/infer.py:22` on MI355X.

Same constructor (a `network.yml` dict), same public methods, attributes, argument meaning, return
types and error behaviour; the Keras/TensorFlow models behind it are replaced by libovn_hip.so
(hand-written HIP kernels, C ABI in include/ovn_hip.h).  Differences a caller can observe:
  * feature volumes (and their spectra for the correlation head) stay resident in HBM between calls: `infer_multiple`
    never re-uploads the cache the way `infer.py:192-193` rebuilds `np.array(self.feature_volumes)`, and
    `self.feature_volumes` is a list-like VIEW of that device cache that copies a volume to the host only when it is
    indexed (`len()`, `[i]`, iteration, `np.array()` and `.append()` behave like the reference's list);
  * arithmetic: fp32 storage and accumulation everywhere; the contractions run on the fp16 matrix cores with a scaled
    3-term split ("f16x3": 22 significand bits per operand -- measured against an fp64 evaluation this is as accurate as
    running every contraction in fp32, profiles/r2_parity_1024.json).  `config['precision'] = 'f32'` (optional key, absent
    from the reference's network.yml) selects bit-for-bit fp32 FMA chains on the fp32 matrix cores at ~1/3 of the speed;
  * `pretrained_weightsfilename` may name a native `.npz` (keys `<layer>/kernel|bias`) besides the
    Keras HDF5 file (read by the built-in `hdf5_lite` parser);
  * `infer_best_match` (extension): `infer_multiple` + demo3's decision taken on the GPU;
  * `config['scan_folder']` (extension key): read the RAW scans `<scan_folder>/<frame>.bin` and project them on the GPU
    (`ovn_project`: range image + normals + intensity straight into the stacked leg input) instead of reading demo1's `.npy`
    files -- demo1 + demo2/demo3 in one object (BASELINE configs[4]); same bits as the `.npy` route on files written by
    `overlapnet_amd.preprocess`.  The projection reproduces the reference's `utils.py` as it runs under NumPy >= 1.22 on an
    AVX512_SKX x86-64 host (float32 `arctan2` / `arcsin` = Intel SVML, bit for bit); `config['projection_trig'] = 'rounded'`
    selects the correctly rounded float32 angles instead (what NumPy's libm fallback gives on AVX2-only / aarch64 hosts: a point
    moves to the neighbouring pixel about once per 200 k points);
  * `self.leg` / `self.head` are the native engine, not keras.Model objects.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import weights as W
from ._lib import OvnError
from .engine import FEAT_C, FEAT_W, OvnEngine

_VALID_LEGS = ("360OutputkLegs", "360OutputkLegsFixed")       # generateNet.py:119,222


class FeatureVolumeCache(object):
  """`Infer.feature_volumes`: behaves like the reference's Python list of (1, 360, 128) arrays (infer.py:114,185), but the
  volumes live in HBM (together with their spectra) and are copied to the host one at a time when indexed."""

  def __init__(self, engine, min_capacity=1024, with_delta_cache=True):
    """min_capacity: smallest allocation (volumes) on first use -- 1024 (580 MB with the spectra and the Delta cache rows) for the persistent cache of
    `infer_multiple`, which grows by one frame per call; the throw-away caches of `infer_one` / `infer_multiple_vs_multiple`
    pass the number of volumes they will hold.  with_delta_cache=False: no Delta cache rows (they pay when a cached frame meets
    many queries; a throw-away cache prepares its pairs in the head kernels' scratch instead -- same bits)."""
    self._engine = engine
    self._want_dc = bool(with_delta_cache)
    self._fv = None        # (capacity, 360, 128) device tensor
    self._spec = None      # (capacity, 128, 368) device tensor: cached spectra for the correlation head
    self._dc = None        # (capacity, 49216) device tensor: Delta cache rows (candidate-side half of the Delta head's preparation)
    self._n = 0
    self._min_capacity = max(1, int(min_capacity))

  @property
  def _with_dc(self) -> bool:      # (the engine learns its head geometry when the weights are loaded, after this object is built)
    return self._want_dc and self._engine.has_delta_cache

  # -- device side ---------------------------------------------------------------------------------
  def _grow(self, need: int) -> None:
    if self._fv is not None and need <= self._fv.shape[0]:
      return
    cap = max(self._min_capacity, need if self._fv is None else 2 * need)
    dev = self._engine.device
    nf = torch.empty((cap, FEAT_W, FEAT_C), dtype=torch.float32, device=dev)
    ns = torch.empty((cap, FEAT_C, self._engine.SPEC_W), dtype=torch.float32, device=dev)
    nd = torch.empty((cap if self._with_dc else 0, self._engine.DELTA_CACHE_ELEMS), dtype=torch.float32, device=dev)
    if self._n:
      nf[:self._n].copy_(self._fv[:self._n])
      ns[:self._n].copy_(self._spec[:self._n])
      if self._with_dc:
        nd[:self._n].copy_(self._dc[:self._n])
    self._fv, self._spec, self._dc = nf, ns, nd

  def put_device(self, slot: int, fv: torch.Tensor, spec: Optional[torch.Tensor] = None, dc: Optional[torch.Tensor] = None) -> None:
    """Write k volumes at slots slot .. slot + k - 1 (slot <= len(self): an append when equal, an overwrite below, and -- the
    sharded cache, whose slots fill out of order (distributed.frame_slot) -- beyond the end: `len()` then becomes the high-water
    mark and the slots in between stay unwritten until their frames arrive; nothing refers to a slot before its frame has been fed).
    Spectra / Delta cache rows are computed here unless the caller already has them (the streaming path computes them beside the
    previous frame's head kernels)."""
    k = fv.shape[0]
    slot = int(slot)
    self._grow(max(self._n, slot + k))
    self._fv[slot:slot + k].copy_(fv)
    if spec is not None:
      self._spec[slot:slot + k].copy_(spec)
    else:
      self._engine.spectrum(self._fv[slot:slot + k], out=self._spec[slot:slot + k])
    if self._with_dc:
      if dc is not None:
        self._dc[slot:slot + k].copy_(dc)
      else:
        self._engine.delta_cache(self._fv[slot:slot + k], out=self._dc[slot:slot + k])
    self._n = max(self._n, slot + k)

  def put_slots_device(self, slots, fv: torch.Tensor) -> None:
    """Write k volumes at k arbitrary (distinct) slots: ONE batched spectrum / Delta-row computation on the contiguous input and one
    indexed copy per pool -- the sharded cache rebuilt from a list, whose slots are not consecutive (distributed.frame_slot)."""
    slots = np.asarray(slots, dtype=np.int64).reshape(-1)
    k = fv.shape[0]
    if k == 0:
      return
    if len(slots) != k or len(np.unique(slots)) != k or slots.min() < 0:
      raise ValueError('put_slots_device: %d volumes for slots %s' % (k, slots[:8]))
    self._grow(max(self._n, int(slots.max()) + 1))
    idx = torch.from_numpy(slots).to(fv.device)
    self._fv.index_copy_(0, idx, fv)
    self._spec.index_copy_(0, idx, self._engine.spectrum(fv))
    if self._with_dc:
      self._dc.index_copy_(0, idx, self._engine.delta_cache(fv))
    self._n = max(self._n, int(slots.max()) + 1)

  def extend_device(self, fv: torch.Tensor, spec: Optional[torch.Tensor] = None, dc: Optional[torch.Tensor] = None) -> None:
    """Append k volumes (see put_device)."""
    self.put_device(self._n, fv, spec=spec, dc=dc)

  @property
  def device_features(self) -> torch.Tensor:
    return self._fv[:self._n] if self._fv is not None else torch.empty((0, FEAT_W, FEAT_C), device=self._engine.device)

  @property
  def device_spectra(self) -> torch.Tensor:
    return self._spec[:self._n] if self._spec is not None else torch.empty((0, FEAT_C, self._engine.SPEC_W), device=self._engine.device)

  @property
  def device_delta_cache(self) -> Optional[torch.Tensor]:
    """Delta cache rows of the cached volumes, or None when the head geometry has none (conv1size != 15)."""
    if not self._with_dc:
      return None
    return self._dc[:self._n] if self._dc is not None else torch.empty((0, self._engine.DELTA_CACHE_ELEMS), device=self._engine.device)

  # -- list / ndarray behaviour ----------------------------------------------------------------------
  def __len__(self):
    return self._n

  @property
  def shape(self):
    return (self._n, 1, FEAT_W, FEAT_C)

  def _one(self, i):
    a = self._fv[i].cpu().numpy()
    return a.reshape(1, FEAT_W, FEAT_C)

  def __getitem__(self, i):
    if isinstance(i, slice):
      return [self._one(j) for j in range(*i.indices(self._n))]
    i = int(i)
    if i < 0:
      i += self._n
    if not 0 <= i < self._n:
      raise IndexError('list index out of range')
    return self._one(i)

  def __iter__(self):
    for i in range(self._n):
      yield self._one(i)

  def __array__(self, dtype=None, copy=None):
    a = self.device_features.cpu().numpy().reshape(self._n, 1, FEAT_W, FEAT_C)
    return a.astype(dtype) if dtype is not None else a

  def append(self, volume) -> None:
    """list.append of a host (1, 360, 128) volume, as a caller of the reference could do."""
    v = torch.from_numpy(np.ascontiguousarray(volume, np.float32).reshape(1, FEAT_W, FEAT_C)).to(self._engine.device)
    self.extend_device(v)

_VALID_OVERLAP_HEADS = ("DeltaLayerConv1NetworkHead",)         # generateNet.py:64
_VALID_ORIENTATION_HEADS = ("CorrelationHead",)                # generateNet.py:327


class Infer():
  """ A class used for inferring overlap and yaw-angle between LiDAR scans (MI355X-native). """

  def __init__(self, config, device: Optional[int] = None, weights: Optional[dict] = None, seed: int = 0,
               rank: Optional[int] = None, world: Optional[int] = None, group=None):
    """ Args:
          config: dict with configuration values, usually loaded from network.yml (infer.py:26-84).
          device / weights / seed: extensions -- HIP device index, an in-memory weight dict that
          overrides `pretrained_weightsfilename`, and the seed of the Keras-default random init that
          is used when no weights are given (the reference keeps Keras' random init, infer.py:121-122).
          rank / world / group: extension -- the 1-vs-N sweep of `infer_multiple` / `infer_best_match` sharded over `world`
          processes (one per GPU, torch.distributed initialised by the caller).  Every rank makes the SAME calls
          with the same arguments and gets the same results.  Frame i's feature volume / spectrum / Delta row live on rank
          `distributed.frame_owner(i)` only (skewed block-cyclic: consecutive frames go to consecutive ranks, and a frame's local
          slot stays congruent to its id modulo 32 -- a gated window of consecutive ids, demo3_lcd.py:92-115, is spread evenly),
          the query leg runs on every rank (cheaper than a broadcast), each rank scores the references it owns and ONE all-gather
          of 8 B per reference (or of one 16-byte best-match record per rank) merges the answer.  Same bits as the unsharded
          object.  `infer_one`, `infer_multiple_vs_multiple` and `create_feature_volumes` run replicated on every rank.
          `len(infer.feature_volumes)` is the rank's LOCAL high-water slot there (not the number of frames fed), and indexing
          the view addresses local slots.  A failure of one rank's local work (a file missing on that node, a kernel error) is
          carried in the collective's payload: every rank raises, none is left waiting in the all-gather.
          config['stream_ahead'] (optional, default True): the speculative read + leg of frame i + 1 beside frame i's heads.
    """
    self._rank, self._world, self._group = 0, 1, group
    if world is not None and int(world) > 1:
      import torch.distributed as dist
      if not dist.is_initialized():
        raise Exception('Infer(world=%d): initialise torch.distributed first (one process per GPU)' % int(world))
      self._world = int(world)
      self._rank = int(dist.get_rank(group) if rank is None else rank)
      if not 0 <= self._rank < self._world or self._world != dist.get_world_size(group):
        raise Exception('Infer: rank %d / world %d do not match the process group' % (self._rank, self._world))
    self._n_frames = 0          # sharded mode: frames fed so far (== the next frame id)
    # sharded mode: what this rank did (tests assert that the cache-row path was taken, not a per-pair fallback)
    self.sharded_stats = {'frames_cached': 0, 'pairs_scored': 0, 'pairs_on_cache_rows': 0, 'ahead_delta_rows': 0}
    self._scan_folder = config.get('scan_folder') or None
    self._projection_trig = config.get('projection_trig', 'numpy_avx512')     # extension key, with 'scan_folder' (engine.set_projection_trig)
    if self._scan_folder is not None and config['use_class_probabilities']:
      raise Exception("config['scan_folder']: the semantic channels come from RangeNet++ .npy files, not from the raw scans")
    self._stream_ahead = bool(config.get('stream_ahead', True))
    self.network_output_size = config['model']['leg_output_width']      # infer.py:31
    self.seq = config['infer_seqs']                                     # infer.py:32
    self.datasetpath = config['data_root_folder']                       # infer.py:34

    self.use_depth = config['use_depth'] if 'use_depth' in config else True
    self.use_normals = config['use_normals'] if 'use_normals' in config else True
    self.use_class_probabilities = config['use_class_probabilities'] if 'use_class_probabilities' in config else False
    self.use_class_probabilities_pca = (config['use_class_probabilities_pca']
                                        if 'use_class_probabilities_pca' in config else False)
    self.use_intensity = config['use_intensity'] if 'use_intensity' in config else False

    # channel count: the reference indexes the config unconditionally here (infer.py:62-73 -> KeyError)
    self.no_input_channels = 0
    if config['use_depth']:
      self.no_input_channels += 1
    if config['use_normals']:
      self.no_input_channels += 3
    if config['use_intensity']:
      self.no_input_channels += 1
    if config['use_class_probabilities']:
      if config['use_class_probabilities_pca']:
        self.no_input_channels += 3
      else:
        self.no_input_channels += 20

    # input shape, mutated in place exactly like infer.py:76-82
    self.inputShape = config['model']['inputShape']
    if len(self.inputShape) == 3:
      pass
    elif len(self.inputShape) == 2:
      self.inputShape.append(self.no_input_channels)
    else:
      self.inputShape[2] = self.no_input_channels

    self.batch_size = config['batch_size']                              # infer.py:84

    # name dispatch of infer.py:87-93 (getattr(generateNet, 'generate' + name) -> AttributeError)
    model_cfg = config['model']
    legsType = model_cfg['legsType']
    overlap_head = model_cfg['overlap_head']
    orientation_head = model_cfg['orientation_head']
    for name, valid in ((legsType, _VALID_LEGS), (overlap_head, _VALID_OVERLAP_HEADS),
                        (orientation_head, _VALID_ORIENTATION_HEADS)):
      if name not in valid:
        raise AttributeError("module 'generateNet' has no attribute 'generate%s'" % name)
    if self.network_output_size != FEAT_W:
      raise OvnError("leg_output_width=%s: the HIP heads are built for 360" % self.network_output_size)

    self._model_cfg = {
      'strides_layer1': model_cfg.get('strides_layer1', (2, 2)),
      'additional_unsymmetric_layer3a': model_cfg.get('additional_unsymmetric_layer3a', False),
      'conv1NetworkHead_conv1size': model_cfg.get('conv1NetworkHead_conv1size', 15),
    }
    self.engine = OvnEngine(self.inputShape[0], self.inputShape[1], self.inputShape[2], device=device)
    self.leg = self.engine    # reference: keras.Model (infer.py:101)
    self.head = self.engine   # reference: keras.Model (infer.py:111)
    self.precision = config.get('precision', 'f16x3')   # extension key, see the module docstring
    if self.precision not in ('f16x3', 'f32'):
      raise Exception("config['precision'] must be 'f16x3' or 'f32'")
    self.engine.set_leg_precision(self.precision)
    self.engine.set_head_precision(self.precision)
    self.engine.set_projection_trig(self._projection_trig)

    # previous feature volumes (infer.py:114): list-like view of the HBM-resident cache
    self._feature_volumes = FeatureVolumeCache(self.engine)

    pretrained_weightsfilename = config['pretrained_weightsfilename']
    if weights is not None:
      w = weights
    elif len(pretrained_weightsfilename) > 0:
      w = W.load_weights_file(pretrained_weightsfilename)
    else:
      print('Pre-trained weights was not found in:', pretrained_weightsfilename)
      w = W.keras_default_init(self.no_input_channels, self._model_cfg, seed)
    self.engine.load_weights(w, self._model_cfg)
    self._weights = w           # kept for the second context of the streaming path (_start_ahead)
    self._qa = None
    self._ahead_fv = None       # key of the frame whose leg is running (or done) on the side stream
    self._ahead_sig = None      # (datasetpath, seq, name, (mtime_ns, size) of every cue file) the speculative read saw

  def close(self) -> None:
    """Release the second (look-ahead) context and the engine; the object cannot be used afterwards."""
    if self._qa is not None:
      try:
        self._qa.close()
      finally:
        self._qa = None
    if getattr(self, 'engine', None) is not None:
      self.engine.close()

  def __del__(self):
    try:
      if getattr(self, '_qa', None) is not None:
        self._qa.close()
        self._qa = None
    except Exception:
      pass

  @property
  def feature_volumes(self):
    """The cache of previous feature volumes (infer.py:114,185,220), resident in HBM."""
    return self._feature_volumes

  @feature_volumes.setter
  def feature_volumes(self, value):
    """`infer.feature_volumes = []` (the reference's way to reset the cache) or any list / ndarray of (1, 360, 128) host volumes
    rebuilds the device cache from it; a FeatureVolumeCache is taken as is."""
    if isinstance(value, FeatureVolumeCache):
      self._drop_ahead()
      self._feature_volumes = value
      if self._world > 1:
        # a cache built elsewhere does not follow this object's frame ownership (slot = distributed.frame_slot): the sharded
        # sweeps refuse it until the cache is reset from a list (`infer.feature_volumes = []` or the volumes of frames 0 .. n-1)
        self._n_frames = -1
      return
    vols = np.asarray(value, dtype=np.float32) if len(value) else np.zeros((0, FEAT_W, FEAT_C), np.float32)
    if vols.size % (FEAT_W * FEAT_C):
      raise ValueError('feature volumes must have shape (n, 1, %d, %d)' % (FEAT_W, FEAT_C))
    vols = np.ascontiguousarray(vols).reshape(-1, FEAT_W, FEAT_C)
    self._drop_ahead()
    self._n_frames = vols.shape[0]          # sharded mode: the list holds the volumes of frames 0 .. n-1; the next frame is n
    cache = FeatureVolumeCache(self.engine)
    if self._world > 1 and vols.shape[0]:   # ... of which this rank keeps the ones it owns, each at its slot
      from . import distributed as D
      ids = np.arange(vols.shape[0])
      mine = ids[D.frame_owner(ids, self._world) == self._rank]
      slots = D.frame_slot(mine, self._world)
      # ONE host-to-device copy of everything this rank owns, ONE batched spectrum and Delta-row computation, one indexed copy per
      # pool (under the skewed block-cyclic ownership a rank's slots are hardly ever consecutive: per-frame writes were one copy and
      # two launches per frame -- ADVICE r5)
      if len(mine):
        dev_vols = torch.from_numpy(np.ascontiguousarray(vols[mine])).to(self.engine.device)
        for a in range(0, len(mine), 4096):      # (chunks bound the transient spectra / Delta rows)
          cache.put_slots_device(slots[a:a + 4096], dev_vols[a:a + 4096])
    elif vols.shape[0]:
      cache.extend_device(torch.from_numpy(vols).to(self.engine.device))
    self._feature_volumes = cache

  # ------------------------------------------------------------------------------------------------
  # ---- inputs: channel stacking of ImagePairOverlapOrientationSequence.prepareOneInput (:130-207), depth -> normals -> class
  #      probabilities -> intensity, raw values; one pinned, contiguous host buffer per cue, interleaved on the GPU ----
  def _cue_files(self):
    """(sub-folder, channels, error label or None) per cue in the reference's channel order (:143-207)."""
    cues = []
    if self.use_depth:
      cues.append(('depth', 1, 'depth'))
    if self.use_normals:
      cues.append(('normal', 3, 'normal'))
    if self.use_class_probabilities:
      cues.append(('probability_pca', 3, None) if self.use_class_probabilities_pca else ('probability', 20, None))
    if self.use_intensity:
      cues.append(('intensity', 1, None))
    return cues

  @staticmethod
  def _read_npy_into(path: str, dst: np.ndarray) -> None:
    """np.load(path) into `dst` without the intermediate array when the file holds exactly dst's dtype / shape (C order);
    raises IOError like np.load when the file cannot be opened."""
    with open(path, 'rb') as f:
      try:
        version = np.lib.format.read_magic(f)
        shape, fortran, dtype = (np.lib.format.read_array_header_1_0(f) if version == (1, 0)
                                 else np.lib.format.read_array_header_2_0(f))
      except Exception:
        shape = None
      if shape is not None and not fortran and dtype == dst.dtype and tuple(shape) == tuple(dst.shape) and dst.flags['C_CONTIGUOUS']:
        if f.readinto(memoryview(dst).cast('B')) == dst.nbytes:
          return
    dst[...] = np.load(path)

  def _ensure_stage(self, n: int) -> None:
    """Two sets of pinned staging buffers used alternately: a set is free again once ITS host-to-device copies are done (an event
    recorded right behind them), not when the leg that consumes the device copy has finished -- the host never waits for the GPU."""
    h, w, c = self.inputShape
    if getattr(self, '_stage', None) is None or self._stage_n < n:
      self._stage_n = max(n, 1)
      self._stage = [{sub: torch.empty((self._stage_n, h, w) + ((k,) if k > 1 else ()), dtype=torch.float32).pin_memory()
                      for sub, k, _ in self._cue_files()} for _ in range(2)]
      self._stage_free = [None, None]
      self._stage_next = 0
      self._ahead = None

  def _read_cues(self, slot: int, filenames: Sequence[str]) -> None:
    """Every cue's files of `filenames` straight into staging set `slot` (depth (n,h,w), normals (n,h,w,3), ...)."""
    root = os.path.join(self.datasetpath, self.seq)
    if self._stage_free[slot] is not None:
      self._stage_free[slot].synchronize()
    for sub, k, label in self._cue_files():
      hv = self._stage[slot][sub][:len(filenames)].numpy()
      for i, name in enumerate(filenames):
        f = os.path.join(root, sub, name + '.npy')
        try:
          self._read_npy_into(f, hv[i])
        except IOError:
          if label is not None:
            raise Exception('Could not read %s image %s' % (label, f))
          hv[i] = np.load(os.path.join(root, sub, name + '.npz'))

  def _readahead(self, filenames: Sequence[str]) -> None:
    """Read the files the NEXT call will most likely ask for into the staging set that call will use, while the GPU is busy with the
    current one (a streaming loop-closure run asks for frame i + 1 after frame i; the reads of one frame cost ~0.15 ms of host time
    that would otherwise sit between two frames with the GPU idle).  Purely speculative: any failure, or a different request, and
    the next call reads its files as usual."""
    try:
      self._ensure_stage(len(filenames))
      slot = self._stage_next
      self._ahead = None
      self._read_cues(slot, filenames)
      self._ahead = (tuple(filenames), slot)
    except Exception:
      self._ahead = None

  def _scan_path(self, name: str) -> str:
    return os.path.join(self._scan_folder, name + '.bin')

  def _inputs_from_scans(self, filenames: Sequence[str], engine=None) -> torch.Tensor:
    """(n,h,w,C) leg input from the RAW scans (gen_depth_data.py:31-32 reads them the same way): one batched `ovn_project`
    (utils.py:59-186 on the GPU) writes depth | normals | intensity in the reference's channel order straight into the stacked
    tensor.  `engine`: the context whose scratch the projection uses (the look-ahead passes its second context)."""
    eng = engine or self.engine
    pts, offs = [], [0]
    for name in filenames:
      f = self._scan_path(name)
      try:
        p = np.fromfile(f, dtype=np.float32)
      except (IOError, OSError):
        raise Exception('Could not read scan file %s' % f)
      p = p.reshape((-1, 4))
      pts.append(p)
      offs.append(offs[-1] + p.shape[0])
    flat = np.concatenate(pts, axis=0) if offs[-1] else np.zeros((1, 4), np.float32)
    dev = eng.device
    pd = torch.from_numpy(np.ascontiguousarray(flat)).to(dev, non_blocking=False)
    od = torch.tensor(offs, dtype=torch.int64, device=dev)
    h, w, c = self.inputShape
    r = eng.project(pd, od, max(p.shape[0] for p in pts) if pts else 0, h, w, want=(),
                    stacked_flags=(bool(self.use_depth), bool(self.use_normals), bool(self.use_intensity)))
    return r["stacked"]

  def _inputs_device(self, filenames: Sequence[str], engine=None) -> torch.Tensor:
    """(n,h,w,C) leg input on the device: every cue's files are read straight into a pinned staging buffer, copied asynchronously
    and interleaved by one device-side concatenation (a strided host-side interleave plus a pageable copy cost more than the leg
    itself for a single frame)."""
    n = len(filenames)
    dev = self.engine.device
    if self._scan_folder is not None:
      return self._inputs_from_scans(filenames, engine)
    self._ensure_stage(n)
    slot = self._stage_next
    self._stage_next ^= 1
    if self._ahead != (tuple(filenames), slot):
      self._read_cues(slot, filenames)
    self._ahead = None
    parts = []
    for sub, k, label in self._cue_files():
      d = self._stage[slot][sub][:n].to(dev, non_blocking=True)
      parts.append(d if k > 1 else d.unsqueeze(-1))
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    self._stage_free[slot] = ev
    x = parts[0].contiguous() if len(parts) == 1 else torch.cat(parts, dim=-1)
    return x

  def _leg_device(self, filenames: Sequence[str]) -> torch.Tensor:
    """leg over `filenames`, batched like predict_generator (batch_size scans per launch group)."""
    n = len(filenames)
    out = torch.empty((n, FEAT_W, FEAT_C), dtype=torch.float32, device=self.engine.device)
    bs = max(1, int(self.batch_size))
    for s in range(0, n, bs):
      k = min(bs, n - s)
      x = self._inputs_device(filenames[s:s + k])
      self.engine.leg(x, out=out[s:s + k])
    return out

  def create_feature_volumes(self, filenames):
    """ create feature volumes, thus execute the leg (infer.py:240-265).
        Returns a n x 1 x 360 x 128 numpy array. """
    fv = self._leg_device(list(filenames))
    return fv.cpu().numpy().reshape(len(filenames), 1, FEAT_W, FEAT_C)

  # ------------------------------------------------------------------------------------------------
  def _heads_device(self, cache: FeatureVolumeCache, pair_indizes: np.ndarray):
    """pairs[:,0] -> head-left, pairs[:,1] -> head-right (ImagePairOverlapSequenceFeatureVolume.py:43-47).
    Delta head on the cached feature volumes, correlation head in its spectral form on the cached spectra; the pair
    indices are range-checked on the host (no device synchronisation before the launches).  When every pair has the
    same right-hand volume (the 1-vs-N sweep of `infer_multiple`) the library's 1-vs-N form is used: no right index
    array, and the query's linear term is evaluated once instead of per pair."""
    feats, spec = cache.device_features, cache.device_spectra
    right = pair_indizes[:, 1]
    left = pair_indizes[:, 0]
    if len(left) and left[0] == 0 and np.array_equal(left, np.arange(len(left))):
      left = None                                   # candidates 0 .. n-1 in order: no index array to upload
    if len(right) and np.all(right == right[0]):
      q = int(right[0])
      if not 0 <= q < len(cache):
        raise IndexError('index %d is out of bounds for axis 0 with size %d' % (q, len(cache)))
      return self.engine.heads(feats, feats[q:q + 1], lidx=left, n=len(right), spec_l=spec, spec_r=spec[q:q + 1],
                               dcache_l=cache.device_delta_cache)
    return self.engine.heads(feats, feats, lidx=left, ridx=right, n=len(right), spec_l=spec, spec_r=spec)

  def _run_heads(self, cache: FeatureVolumeCache, pair_indizes: np.ndarray, ahead=None):
    r = self._heads_device(cache, pair_indizes)
    if ahead is not None:
      self._start_ahead(ahead)    # next frame: file reads, copy and leg in the shadow of the head kernels just enqueued
    res = torch.stack([r["overlap"].view(torch.int32), r["yaw"]]).cpu().numpy()      # ONE device-to-host copy for both
    overlap = res[0].view(np.float32).reshape(-1, 1)
    yaw = res[1].astype(np.int64)
    return overlap, yaw

  def infer_one(self, filepath1, filepath2):
    """ Infer with one input pair (infer.py:124-160). Returns (overlap (1,) f32, yaw (1,) int). """
    if not filepath1.endswith('.bin') or not filepath2.endswith('.bin'):
      raise Exception('Please check the LiDAR file format, '
                      'this implementation currently only works with .bin files.')
    filename1 = os.path.basename(filepath1).replace('.bin', '')
    filename2 = os.path.basename(filepath2).replace('.bin', '')
    self.filenames = np.array([filename2, filename1])

    preprocess_data_folder = os.path.join(self.datasetpath, self.seq)
    if self._scan_folder is None and not os.path.isdir(preprocess_data_folder):
      raise Exception('Please first generate preprocessed input data.')

    pair = FeatureVolumeCache(self.engine, min_capacity=2, with_delta_cache=False)
    pair.extend_device(self._leg_device(list(self.filenames)))
    indizes = np.zeros((1, 2), dtype=int)
    indizes[0, 0] = 0
    indizes[0, 1] = 1
    overlap, yaw = self._run_heads(pair, indizes)
    return overlap[0], yaw

  # ---- streaming: the NEXT frame's files, host-to-device copy and leg in the shadow of the CURRENT frame's head kernels ----------
  def _start_ahead(self, current_frame_id) -> None:
    """After the head kernels of frame i have been enqueued: read frame i + 1's cue files (host, ~0.15 ms), then copy them and run
    their leg on a second library context and stream (engine.QueryAhead) beside those head kernels -- a streaming loop-closure run
    asks for frame i + 1 next (demo3_lcd.py:88-123), and its single-scan leg (0.15 ms of latency-bound kernels) would otherwise
    sit in front of its own heads with the GPU nearly idle.  Purely speculative: a missing file, or a different next request, and
    the next call computes its frame as usual."""
    if not self._stream_ahead:
      return
    try:
      names = [str(int(current_frame_id) + 1).zfill(6)]
    except (TypeError, ValueError):
      return
    self._drop_ahead()
    sig = self._frame_signature(names[0])     # BEFORE the read: a file rewritten during or after it no longer matches
    if sig is None:
      return
    if self._scan_folder is None:
      self._readahead(names)
      if self._ahead is None:
        return
    try:
      if self._qa is None:
        from .engine import QueryAhead
        self._qa = QueryAhead(self.engine, self._weights, self._model_cfg, with_delta_cache=True)
      with torch.cuda.stream(self._qa.stream):
        # copies + interleave (or the projection of the raw scan, in the second context's scratch) on the side stream
        x = self._inputs_device(names, engine=self._qa.side)
      want_dc = True
      if self._world > 1:     # only the frame's owner keeps (and therefore computes) its Delta cache row
        from . import distributed as D
        want_dc = bool(D.frame_owner(int(current_frame_id) + 1, self._world) == self._rank)
        self.sharded_stats['ahead_delta_rows'] += int(want_dc)
      self._qa.submit(x, wait_current=False, with_delta=want_dc)
      self._ahead_fv = names[0]
      self._ahead_sig = sig
    except Exception:
      self._ahead_fv = None

  def _frame_signature(self, name: str):
    """(dataset path, sequence, frame name, (mtime_ns, size) of each cue file) -- what a speculative result is keyed by: the look-
    ahead is adopted only if the files are still the ones it read (live preprocessing may rewrite frame i + 1 between two calls;
    `self.seq` / `self.datasetpath` may change).  None if a file is missing."""
    if self._scan_folder is not None:
      try:
        a = os.stat(self._scan_path(name))
      except OSError:
        return None
      return (self._scan_folder, self.seq, name, ((a.st_mtime_ns, a.st_size),))
    root = os.path.join(self.datasetpath, self.seq)
    st = []
    for sub, k, label in self._cue_files():
      f = os.path.join(root, sub, name + '.npy')
      try:
        a = os.stat(f)
      except OSError:
        if label is not None:
          return None
        try:
          a = os.stat(os.path.join(root, sub, name + '.npz'))
        except OSError:
          return None
      st.append((a.st_mtime_ns, a.st_size))
    return (self.datasetpath, self.seq, name, tuple(st))

  def _drop_ahead(self) -> None:
    if self._ahead_fv is not None:
      self._qa.take()             # keeps the helper's in-flight count right; the result is ignored
      self._ahead_fv = None

  def _cache_frame(self, name: str) -> None:
    """Append frame `name` to the feature-volume cache on the current stream: the side stream's feature volume, spectrum and Delta
    cache row if that is the frame it was given, a fresh leg (+ spectrum + row) otherwise -- the same kernels, the same bits."""
    if self._ahead_fv == name and self._ahead_sig is not None and self._ahead_sig == self._frame_signature(name):
      self._ahead_fv = None
      fv, spec, dc = self._qa.take_all()
      self.feature_volumes.extend_device(fv, spec=spec, dc=dc)
      return
    self._drop_ahead()
    self._ahead = None          # a stale speculative read of the staging buffers is not adopted either
    self.feature_volumes.extend_device(self._leg_device([name]))

  # ---- sharded 1-vs-N sweep (extension) ------------------------------------------------------------------------------
  def _check_sharded_frame(self, current_frame_id) -> int:
    """Argument checks every rank evaluates identically (they raise on all ranks or on none)."""
    fid = int(current_frame_id)
    if self._n_frames < 0:
      raise Exception('sharded Infer: the cache was replaced by one that does not follow the frame ownership '
                      '(infer_multiple_vs_multiple, or a FeatureVolumeCache assigned to infer.feature_volumes); reset it '
                      '(infer.feature_volumes = []) before the next sharded sweep')
    if fid != self._n_frames:
      raise Exception('sharded Infer: frames must be fed in order 0, 1, 2, ... (the cache index is the frame id, infer.py:184-190); '
                      'got frame %d, expected %d' % (fid, self._n_frames))
    return fid

  def _query_frame_sharded(self, fid: int):
    """Leg (+ spectrum; the Delta row on the owner only) of the current frame on EVERY rank; the owner writes it to the frame's
    slot of its local cache.  Returns the query's (feature volume, spectrum) device tensors, valid for the head launches of this
    call."""
    from . import distributed as D
    name = str(fid).zfill(6)
    owned = D.frame_owner(fid, self._world) == self._rank
    if self._ahead_fv == name and self._ahead_sig is not None and self._ahead_sig == self._frame_signature(name):
      self._ahead_fv = None
      fv, spec, dc = self._qa.take_all()
    else:
      self._drop_ahead()
      self._ahead = None
      fv = self._leg_device([name])
      spec = self.engine.spectrum(fv)
      dc = None
    if owned:
      cache = self.feature_volumes
      k = int(D.frame_slot(fid, self._world))
      cache.put_device(k, fv, spec=spec, dc=dc)
      self.sharded_stats['frames_cached'] += 1
      return cache.device_features[k:k + 1], cache.device_spectra[k:k + 1]
    return fv, spec

  def _local_heads_sharded(self, ref: np.ndarray, mine: np.ndarray, q_fv, q_spec):
    """Heads of the references this rank owns, in list order, on their cache rows -> result dict or None."""
    from . import distributed as D
    if not mine.any():
      return None
    cache = self.feature_volumes
    lidx = np.ascontiguousarray(D.frame_slot(ref[mine], self._world), dtype=np.int32)
    if int(lidx.max()) >= len(cache):
      raise OvnError('sharded Infer: reference slot %d beyond the local cache (%d)' % (int(lidx.max()), len(cache)))
    dcl = cache.device_delta_cache
    r = self.engine.heads(cache.device_features, q_fv, lidx=lidx, n=len(lidx), spec_l=cache.device_spectra, spec_r=q_spec,
                          dcache_l=dcl)
    self.sharded_stats['pairs_scored'] += int(len(lidx))
    if dcl is not None:
      self.sharded_stats['pairs_on_cache_rows'] += int(len(lidx))      # offered to the library with their Delta cache rows
    return r

  def _sharded_refs(self, reference_frame_id):
    """(reference ids, owner of each, this rank's mask) -- host arithmetic every rank evaluates identically."""
    from . import distributed as D
    ref = np.asarray(reference_frame_id, dtype=np.int64).reshape(-1)
    if len(ref) and (ref.min() < 0 or ref.max() > self._n_frames):     # (the current frame, id == _n_frames, is cached by this call)
      raise IndexError('index %d is out of bounds for axis 0 with size %d' % (int(ref.max()), self._n_frames + 1))
    owner = D.frame_owner(ref, self._world) if len(ref) else np.zeros(0, np.int64)
    return ref, owner, owner == self._rank

  @staticmethod
  def _raise_rank_failure(statuses, local_error):
    bad = [int(r) for r in np.nonzero(np.asarray(statuses) != 0)[0]]
    if local_error is not None:
      raise local_error
    if bad:
      raise Exception('sharded Infer: the local work of rank(s) %s failed (see their own exception); the query is void on every rank' % bad)

  def _frame_done(self, statuses, local_error):
    """After the status exchange of a sharded call: the frame counts as fed -- `_n_frames` advances, on EVERY rank -- only when no
    rank's local work failed.  Otherwise every rank raises and the frame id stays the next expected one: the owner's slot may be
    unwritten (its leg failed), so nothing may refer to it; the caller repairs the cause and feeds the SAME frame again
    (`put_device` overwrites the slot where it was written)."""
    if local_error is None and not np.any(np.asarray(statuses) != 0):
      self._n_frames += 1
      return
    self._drop_ahead()
    self._ahead = None
    self._raise_rank_failure(statuses, local_error)

  def _agree_frame_cached(self, local_error):
    """A sharded call with an EMPTY reference list (the first frames of demo3) has no scores to exchange, but the ranks must still
    agree on whether the current frame was cached: one 16-byte record per rank, word 3 < 0 = this rank's local work failed."""
    from . import distributed as D
    rec = torch.tensor([-1, 0, 0, -1 if local_error is not None else 0], dtype=torch.int32, device=self.engine.device)
    recs = D.allgather_records(rec, self._group)
    self._frame_done((recs.reshape(-1, 4)[:, 3] < 0).numpy(), local_error)

  def _infer_multiple_sharded(self, current_frame_id, reference_frame_id):
    from . import distributed as D
    fid = self._check_sharded_frame(current_frame_id)
    ref, owner, mine = self._sharded_refs(reference_frame_id)
    dev = self.engine.device
    err, r = None, None
    try:                                      # local work: files, leg, heads -- may fail on ONE rank only
      q_fv, q_spec = self._query_frame_sharded(fid)
      if len(ref):
        r = self._local_heads_sharded(ref, mine, q_fv, q_spec)
    except Exception as e:                    # noqa: BLE001 -- carried to every rank in the payload below
      err = e
    if len(ref) == 0:                         # no scores to exchange: the ranks still agree on whether the frame was cached
      self._agree_frame_cached(err)
      if err is None:
        self._start_ahead(current_frame_id)
      return None
    if err is None:
      self._start_ahead(current_frame_id)
    ov = r["overlap"] if r is not None else torch.empty(0, dtype=torch.float32, device=dev)
    yw = r["yaw"] if r is not None else torch.empty(0, dtype=torch.int32, device=dev)
    ov_all, yaw_all, statuses = D.allgather_by_owner(ov, yw, owner, self._group, status=0 if err is None else 1)
    self._frame_done(statuses, err)
    res = torch.stack([ov_all.view(torch.int32), yaw_all]).cpu().numpy()      # ONE device-to-host copy for both
    overlap = res[0].view(np.float32).reshape(-1, 1)
    return overlap.squeeze(), res[1].astype(np.int64)

  def _infer_best_match_sharded(self, current_frame_id, reference_frame_id, overlap_thres):
    from . import distributed as D
    from .engine import decode_match
    fid = self._check_sharded_frame(current_frame_id)
    ref, owner, mine = self._sharded_refs(reference_frame_id)
    err, rec = None, None
    try:
      q_fv, q_spec = self._query_frame_sharded(fid)
      if len(ref):
        r = self._local_heads_sharded(ref, mine, q_fv, q_spec)
        if r is not None:   # the record's id field carries the POSITION in the reference list: ties resolve like np.argmax over the list
          pos = torch.from_numpy(np.nonzero(mine)[0].astype(np.int32)).to(self.engine.device)
          rec = self.engine.best_match(r["overlap"], r["yaw"], overlap_thres, ids=pos)
    except Exception as e:                    # noqa: BLE001
      err = e
    if len(ref) == 0:
      self._agree_frame_cached(err)
      if err is None:
        self._start_ahead(current_frame_id)
      return None
    if err is not None:
      rec = torch.tensor([-1, 0, 0, -1], dtype=torch.int32, device=self.engine.device)      # word 3 < 0: this rank failed
    elif rec is None:
      rec = torch.tensor([-1, 0, 0, 0], dtype=torch.int32, device=self.engine.device)
    if err is None:
      self._start_ahead(current_frame_id)
    recs = D.allgather_records(rec, self._group)
    self._frame_done((recs.reshape(-1, 4)[:, 3] < 0).numpy(), err)
    got = decode_match(D.merge_matches_by_position(recs))
    if got is None:
      return None
    return int(ref[got[0]]), got[1], got[2]

  def infer_multiple(self, current_frame_id, reference_frame_id):
    """ Loop closing: current frame vs old frames (infer.py:162-203).  The current frame's feature
        volume is computed and appended (index == frame id); older ones must already be cached. """
    if self._world > 1:
      return self._infer_multiple_sharded(current_frame_id, reference_frame_id)
    self._cache_frame(str(current_frame_id).zfill(6))

    if len(reference_frame_id) > 0:
      pair_indizes = np.zeros((len(reference_frame_id), 2), dtype=int)
      pair_indizes[:, 1] = np.ones(len(reference_frame_id)) * current_frame_id
      pair_indizes[:, 0] = reference_frame_id
      n = len(self.feature_volumes)
      if pair_indizes.min() < 0 or pair_indizes.max() >= n:
        raise IndexError('index %d is out of bounds for axis 0 with size %d' % (int(pair_indizes.max()), n))
      overlap, yaw = self._run_heads(self.feature_volumes, pair_indizes, ahead=current_frame_id)
      return overlap.squeeze(), yaw
    else:
      return None

  def infer_best_match(self, current_frame_id, reference_frame_id, overlap_thres=0.3):
    """ Addition to the reference API: `infer_multiple` + the loop-closure decision of demo3
        (demo3_lcd.py:117-120) taken on the GPU, so only one record crosses PCIe instead of N scores.
        Returns (reference frame id, overlap, yaw) or None; caches the current frame like `infer_multiple`. """
    from .engine import decode_match
    if self._world > 1:
      return self._infer_best_match_sharded(current_frame_id, reference_frame_id, overlap_thres)
    self._cache_frame(str(current_frame_id).zfill(6))
    if len(reference_frame_id) == 0:
      return None
    ref = np.asarray(reference_frame_id, dtype=np.int64).reshape(-1)
    n = len(self.feature_volumes)
    if ref.min() < 0 or max(int(ref.max()), int(current_frame_id)) >= n:
      raise IndexError('index %d is out of bounds for axis 0 with size %d' % (int(ref.max()), n))
    pair_indizes = np.zeros((len(ref), 2), dtype=np.int64)
    pair_indizes[:, 0] = ref
    pair_indizes[:, 1] = int(current_frame_id)
    r = self._heads_device(self.feature_volumes, pair_indizes)
    ids = torch.from_numpy(ref.astype(np.int32)).to(self.engine.device)
    rec = self.engine.best_match(r["overlap"], r["yaw"], overlap_thres, ids=ids)
    self._start_ahead(current_frame_id)     # next frame's files, copy and leg in the shadow of the kernels just enqueued
    return decode_match(rec)

  def infer_multiple_vs_multiple(self, file_names, first_idxs, second_idxs):
    """ Multiple pairs (infer.py:205-238): pair i = (file_names[first_idxs[i]], file_names[second_idxs[i]]);
        second -> head-left, first -> head-right.  Replaces the feature-volume cache. """
    if len(first_idxs) != len(second_idxs):
      raise Exception('Please make sure the first_idxs and second_idxs have the same size.')
    file_names = [os.path.basename(v).replace('.bin', '') for v in file_names]
    self.feature_volumes = FeatureVolumeCache(self.engine, min_capacity=len(file_names))
    self.feature_volumes.extend_device(self._leg_device(file_names))
    if self._world > 1:
      # replicated on every rank (the pair list is evaluated whole by each): the cache no longer follows the frame ownership of the
      # sharded sweep -- `infer.feature_volumes = []` before the next infer_multiple / infer_best_match
      self._n_frames = -1

    if len(second_idxs) > 0:
      pair_indizes = np.zeros((len(second_idxs), 2), dtype=int)
      pair_indizes[:, 1] = first_idxs
      pair_indizes[:, 0] = second_idxs
      overlap, yaw = self._run_heads(self.feature_volumes, pair_indizes)
      return overlap.squeeze(), yaw
    else:
      return None
