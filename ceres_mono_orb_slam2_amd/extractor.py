"""Host mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:45-111) over the HIP C ABI.

Same constructor arguments, call operator and getters as the reference class; keypoints come back
as a structured array with cv::KeyPoint's seven fields, descriptors as an (N, 32) uint8 array.
"""
import ctypes as C

import numpy as np

from . import _lib

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


class ORBextractor:
    HARRIS_SCORE, FAST_SCORE = 0, 1      # include/ORBextractor.h:49 (HARRIS_SCORE is dead in the reference)

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0):
        self._L = _lib.load()
        h = C.c_void_p()
        _lib.check(self._L.orbx_create(int(nfeatures), float(scaleFactor), int(nlevels), int(iniThFAST),
                                       int(minThFAST), int(device), C.byref(h)), "orbx_create")
        self._h = h
        self.nfeatures, self.nlevels, self.device = int(nfeatures), int(nlevels), int(device)
        self._scale_factor = float(np.float32(scaleFactor))
        self._tables = [np.zeros(nlevels, np.float32) for _ in range(4)] + [np.zeros(nlevels, np.int32)]
        _lib.check(self._L.orbx_get_tables(self._h, *[_lib.ptr(t) for t in self._tables]), "orbx_get_tables")
        self.max_keypoints = self._L.orbx_max_keypoints(self._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.orbx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- getters (include/ORBextractor.h:63-83), returned by value -------------------------------
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self._scale_factor

    def GetScaleFactors(self):
        return self._tables[0].copy()

    def GetInverseScaleFactors(self):
        return self._tables[1].copy()

    def GetScaleSigmaSquares(self):
        return self._tables[2].copy()

    def GetInverseScaleSigmaSquares(self):
        return self._tables[3].copy()

    @property
    def features_per_level(self):
        return self._tables[4].copy()

    # ---- operator() (src/ORBextractor.cc:1043-1105) ------------------------------------------------
    def __call__(self, image, mask=None):
        """image: (H, W) uint8 host array (CV_8UC1).  mask is ignored, as in the reference.
        Returns (keypoints[KP_DTYPE], descriptors[N,32] uint8); an empty image returns empty outputs."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise TypeError("image must be CV_8UC1 (2-D uint8)")      # the reference asserts (:1050)
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = self.max_keypoints
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        _lib.check(self._L.orbx_extract(self._h, _lib.ptr(image), w, h, image.strides[0], _lib.ptr(kps),
                                        _lib.ptr(desc), cap, C.byref(n)), "orbx_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    # ---- batched, device-resident form --------------------------------------------------------------
    def extract_batch(self, d_imgs, stream=None, out=None):
        """d_imgs: torch uint8 CUDA tensor (B, H, W) (rows contiguous).  Enqueues on `stream`
        (default: torch's current stream) and returns torch tensors (kps[B,cap,7] float32 view,
        desc[B,cap,32] uint8, counts[B] int32) without synchronising."""
        import torch
        assert d_imgs.is_cuda and d_imgs.dtype == torch.uint8 and d_imgs.dim() == 3 and d_imgs.stride(2) == 1
        B, h, w = d_imgs.shape
        cap = self.max_keypoints
        if out is None:
            kps = torch.empty((B, cap, 7), dtype=torch.float32, device=d_imgs.device)
            desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=d_imgs.device)
            counts = torch.empty((B,), dtype=torch.int32, device=d_imgs.device)
        else:
            kps, desc, counts = out
        st = torch.cuda.current_stream(d_imgs.device).cuda_stream if stream is None else stream
        _lib.check(self._L.orbx_extract_batch_device(self._h, _lib.ptr(d_imgs), w, h, d_imgs.stride(1),
                                                     d_imgs.stride(0), B, _lib.ptr(kps), _lib.ptr(desc), cap,
                                                     _lib.ptr(counts), C.c_void_p(st)), "orbx_extract_batch_device")
        return kps, desc, counts

    STAGES = ("pyramid", "fast_cells", "octree", "blur", "describe")

    def set_profiling(self, enable=True):
        _lib.check(self._L.orbx_set_profiling(self._h, int(enable)))

    def set_opencv_variant(self, blur_variant):
        """0: OpenCV <= 3.4.1 GaussianBlur (8-bit taps, default), 1: the ufixedpoint16 path of later versions (orbx_set_opencv_variant)"""
        _lib.check(self._L.orbx_set_opencv_variant(self._h, int(blur_variant)), "orbx_set_opencv_variant")

    def stage_ms(self):
        ms = np.zeros(5, np.float32); n = C.c_int()
        _lib.check(self._L.orbx_get_stage_ms(self._h, _lib.ptr(ms), C.byref(n)))
        return dict(zip(self.STAGES, ms.tolist())), n.value

    # ---- mvImagePyramid (include/ORBextractor.h:85) and stage introspection --------------------------
    def level_image(self, level, frame=0, blurred=False):
        w, h = C.c_int(), C.c_int()
        _lib.check(self._L.orbx_get_level_image(self._h, frame, level, int(blurred), None, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        _lib.check(self._L.orbx_get_level_image(self._h, frame, level, int(blurred), _lib.ptr(out), C.byref(w),
                                                C.byref(h)), "orbx_get_level_image")
        return out

    @property
    def mvImagePyramid(self):
        return [self.level_image(l) for l in range(self.nlevels)]

    def _triples(self, fn, level, frame):
        n = C.c_int()
        _lib.check(fn(self._h, frame, level, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 0), 3), np.int32)
        if n.value > 0:
            _lib.check(fn(self._h, frame, level, _lib.ptr(out), n.value, C.byref(n)))
        return out

    def level_candidates(self, level, frame=0):
        return self._triples(self._L.orbx_get_level_candidates, level, frame)

    def level_selected(self, level, frame=0):
        return self._triples(self._L.orbx_get_level_selected, level, frame)
