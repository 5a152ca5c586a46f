"""MI355X-native ORB front-end + bundle-adjustment back-end (drop-in for the hot path of
b51/ceres_mono_orb_slam2).  The product is the HIP library behind include/orbslam_hip.h; this
package is the host-side mirror of the reference's class interface over that C ABI."""
from .extractor import ORBextractor, KP_DTYPE  # noqa: F401
from .matcher import ORBmatcher  # noqa: F401

__all__ = ["ORBextractor", "ORBmatcher", "KP_DTYPE"]
