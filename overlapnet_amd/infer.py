"""`Infer` -- drop-in for the reference's `src/two_heads/infer.py:22` on MI355X.

Same constructor (a `network.yml` dict), same public methods, attributes, argument meaning, return
types and error behaviour; the Keras/TensorFlow models behind it are replaced by libovn_hip.so
(hand-written HIP kernels, C ABI in include/ovn_hip.h).  Differences a caller can observe:
  * feature volumes additionally stay resident in HBM between calls (`infer_multiple` never
    re-uploads the cache the way `infer.py:192-193` rebuilds `np.array(self.feature_volumes)`);
  * `pretrained_weightsfilename` may name a native `.npz` (keys `<layer>/kernel|bias`) besides the
    Keras HDF5 file (read by the built-in `hdf5_lite` parser);
  * `infer_best_match` (extension): `infer_multiple` + demo3's decision taken on the GPU;
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
_VALID_OVERLAP_HEADS = ("DeltaLayerConv1NetworkHead",)         # generateNet.py:64
_VALID_ORIENTATION_HEADS = ("CorrelationHead",)                # generateNet.py:327


class Infer():
  """ A class used for inferring overlap and yaw-angle between LiDAR scans (MI355X-native). """

  def __init__(self, config, device: Optional[int] = None, weights: Optional[dict] = None, seed: int = 0):
    """ Args:
          config: dict with configuration values, usually loaded from network.yml (infer.py:26-84).
          device / weights / seed: extensions -- HIP device index, an in-memory weight dict that
          overrides `pretrained_weightsfilename`, and the seed of the Keras-default random init that
          is used when no weights are given (the reference keeps Keras' random init, infer.py:121-122).
    """
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

    # previous feature volumes (host list like infer.py:114) + their HBM-resident twin
    self.feature_volumes = []
    self._dev_fv: Optional[torch.Tensor] = None   # (capacity, 360, 128) on device
    self._dev_n = 0

    pretrained_weightsfilename = config['pretrained_weightsfilename']
    if weights is not None:
      w = weights
    elif len(pretrained_weightsfilename) > 0:
      w = W.load_weights_file(pretrained_weightsfilename)
    else:
      print('Pre-trained weights was not found in:', pretrained_weightsfilename)
      w = W.keras_default_init(self.no_input_channels, self._model_cfg, seed)
    self.engine.load_weights(w, self._model_cfg)

  # ------------------------------------------------------------------------------------------------
  def _load_inputs(self, filenames: Sequence[str]) -> np.ndarray:
    """Channel stacking of ImagePairOverlapOrientationSequence.prepareOneInput (:130-207):
    depth -> normals -> class probabilities -> intensity, raw values."""
    h, w, c = self.inputShape
    x = np.zeros((len(filenames), h, w, c), dtype=np.float32)
    root = os.path.join(self.datasetpath, self.seq)
    for i, name in enumerate(filenames):
      ch = 0
      if self.use_depth:
        f = os.path.join(root, 'depth', name + '.npy')
        try:
          img = np.load(f)
        except IOError:
          raise Exception('Could not read depth image %s' % f)
        x[i, :, :, ch] = img
        ch += 1
      if self.use_normals:
        f = os.path.join(root, 'normal', name + '.npy')
        try:
          img = np.load(f)
        except IOError:
          raise Exception('Could not read normal image %s' % f)
        x[i, :, :, ch:ch + 3] = img
        ch += 3
      if self.use_class_probabilities:
        sub, k = ('probability_pca', 3) if self.use_class_probabilities_pca else ('probability', 20)
        f = os.path.join(root, sub, name + '.npy')
        try:
          img = np.load(f)
        except IOError:
          img = np.load(os.path.join(root, sub, name + '.npz'))
        x[i, :, :, ch:ch + k] = img
        ch += k
      if self.use_intensity:
        f = os.path.join(root, 'intensity', name + '.npy')
        try:
          img = np.load(f)
        except IOError:
          img = np.load(os.path.join(root, 'intensity', name + '.npz'))
        x[i, :, :, ch] = img
        ch += 1
    return x

  def _leg_device(self, filenames: Sequence[str]) -> torch.Tensor:
    """leg over `filenames`, batched like predict_generator (batch_size scans per launch group)."""
    n = len(filenames)
    out = torch.empty((n, FEAT_W, FEAT_C), dtype=torch.float32, device=self.engine.device)
    bs = max(1, int(self.batch_size))
    for s in range(0, n, bs):
      x = torch.from_numpy(self._load_inputs(filenames[s:s + bs])).to(self.engine.device)
      self.engine.leg(x, out=out[s:s + x.shape[0]])
    return out

  def create_feature_volumes(self, filenames):
    """ create feature volumes, thus execute the leg (infer.py:240-265).
        Returns a n x 1 x 360 x 128 numpy array. """
    fv = self._leg_device(list(filenames))
    return fv.cpu().numpy().reshape(len(filenames), 1, FEAT_W, FEAT_C)

  # ------------------------------------------------------------------------------------------------
  def _append_device(self, fv: torch.Tensor) -> None:
    k = fv.shape[0]
    if self._dev_fv is None or self._dev_n + k > self._dev_fv.shape[0]:
      cap = max(1024, 2 * (self._dev_n + k))
      new = torch.empty((cap, FEAT_W, FEAT_C), dtype=torch.float32, device=self.engine.device)
      if self._dev_fv is not None and self._dev_n:
        new[:self._dev_n].copy_(self._dev_fv[:self._dev_n])
      self._dev_fv = new
    self._dev_fv[self._dev_n:self._dev_n + k].copy_(fv)
    self._dev_n += k

  def _run_heads(self, feats: torch.Tensor, pair_indizes: np.ndarray):
    """pairs[:,0] -> head-left, pairs[:,1] -> head-right (ImagePairOverlapSequenceFeatureVolume.py:43-47)."""
    r = self.engine.heads(feats, feats, lidx=pair_indizes[:, 0], ridx=pair_indizes[:, 1])
    overlap = r["overlap"].cpu().numpy().reshape(-1, 1)
    yaw = r["yaw"].cpu().numpy().astype(np.int64)
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
    if not os.path.isdir(preprocess_data_folder):
      raise Exception('Please first generate preprocessed input data.')

    fv = self._leg_device(list(self.filenames))
    indizes = np.zeros((1, 2), dtype=int)
    indizes[0, 0] = 0
    indizes[0, 1] = 1
    overlap, yaw = self._run_heads(fv, indizes)
    return overlap[0], yaw

  def infer_multiple(self, current_frame_id, reference_frame_id):
    """ Loop closing: current frame vs old frames (infer.py:162-203).  The current frame's feature
        volume is computed and appended (index == frame id); older ones must already be cached. """
    filename = [str(current_frame_id).zfill(6)]
    fv = self._leg_device(filename)
    self.feature_volumes.append(fv[0].cpu().numpy().reshape(1, FEAT_W, FEAT_C))
    self._append_device(fv)

    if len(reference_frame_id) > 0:
      pair_indizes = np.zeros((len(reference_frame_id), 2), dtype=int)
      pair_indizes[:, 1] = np.ones(len(reference_frame_id)) * current_frame_id
      pair_indizes[:, 0] = reference_frame_id
      if pair_indizes.min() < 0 or pair_indizes.max() >= self._dev_n:
        raise IndexError('index %d is out of bounds for axis 0 with size %d' % (int(pair_indizes.max()), self._dev_n))
      overlap, yaw = self._run_heads(self._dev_fv[:self._dev_n], pair_indizes)
      return overlap.squeeze(), yaw
    else:
      return None

  def infer_best_match(self, current_frame_id, reference_frame_id, overlap_thres=0.3):
    """ Addition to the reference API: `infer_multiple` + the loop-closure decision of demo3
        (demo3_lcd.py:117-120) taken on the GPU, so only one record crosses PCIe instead of N scores.
        Returns (reference frame id, overlap, yaw) or None; caches the current frame like `infer_multiple`. """
    from .engine import decode_match
    filename = [str(current_frame_id).zfill(6)]
    fv = self._leg_device(filename)
    self.feature_volumes.append(fv[0].cpu().numpy().reshape(1, FEAT_W, FEAT_C))
    self._append_device(fv)
    if len(reference_frame_id) == 0:
      return None
    ref = np.asarray(reference_frame_id, dtype=np.int64).reshape(-1)
    if ref.min() < 0 or max(int(ref.max()), int(current_frame_id)) >= self._dev_n:
      raise IndexError('index %d is out of bounds for axis 0 with size %d' % (int(ref.max()), self._dev_n))
    feats = self._dev_fv[:self._dev_n]
    ids = torch.from_numpy(ref.astype(np.int32)).to(self.engine.device)
    r = self.engine.heads(feats, feats, lidx=ids, ridx=np.full(len(ref), int(current_frame_id)))
    return decode_match(self.engine.best_match(r["overlap"], r["yaw"], overlap_thres, ids=ids))

  def infer_multiple_vs_multiple(self, file_names, first_idxs, second_idxs):
    """ Multiple pairs (infer.py:205-238): pair i = (file_names[first_idxs[i]], file_names[second_idxs[i]]);
        second -> head-left, first -> head-right.  Replaces the feature-volume cache. """
    if len(first_idxs) != len(second_idxs):
      raise Exception('Please make sure the first_idxs and second_idxs have the same size.')
    file_names = [os.path.basename(v).replace('.bin', '') for v in file_names]
    fv = self._leg_device(file_names)
    self.feature_volumes = fv.cpu().numpy().reshape(len(file_names), 1, FEAT_W, FEAT_C)
    self._dev_fv = fv
    self._dev_n = fv.shape[0]

    if len(second_idxs) > 0:
      pair_indizes = np.zeros((len(second_idxs), 2), dtype=int)
      pair_indizes[:, 1] = first_idxs
      pair_indizes[:, 0] = second_idxs
      overlap, yaw = self._run_heads(fv, pair_indizes)
      return overlap.squeeze(), yaw
    else:
      return None
