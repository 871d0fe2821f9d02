"""Minimal stand-in for the handful of TensorFlow-1 symbols that the reference drivers examples/example.py (:45, :70-72, :79,
:82-83) and examples/example_v2.py (:50, :76-78, :85, :88-89) touch, so that the unmodified scripts run on the MI355X-native
path without TensorFlow:

    PYTHONPATH=<repo>/python/tf_stub:<repo>/python  MPLBACKEND=Agg  python <reference>/examples/example.py

Put this directory on sys.path ONLY when the real tensorflow is absent.  Nothing here computes anything: the networks
(depthmotionnet.networks_original) run on libdemon_hip.so; `tf.train.Saver().restore` reads the reference's TensorBundle
checkpoint with demon_amd.tf_checkpoint (or a .npz written by demon_amd.weights.save_npz).
"""
import os

__version__ = "1.4.0-demon-amd-stub"


class _Test:
    @staticmethod
    def is_gpu_available(cuda_only=False):
        try:
            from demon_amd import _lib
            return _lib.load().demon_device_count() > 0   # asked through the library itself: no second GPU runtime is loaded
        except Exception:
            return False


test = _Test()


class GPUOptions:
    per_process_gpu_memory_fraction = 1.0


class ConfigProto:
    def __init__(self, allow_soft_placement=False, gpu_options=None, **kwargs):
        self.allow_soft_placement = allow_soft_placement
        self.gpu_options = gpu_options


class InteractiveSession:
    """holds the weights for the network classes (they read `session.demon_weights`)"""

    def __init__(self, config=None, **kwargs):
        self.config = config
        self.demon_weights = None

    def run(self, fetches=None, feed_dict=None):
        return None  # only ever called with global_variables_initializer() by the example

    def close(self):
        pass


Session = InteractiveSession


def global_variables_initializer():
    return None


class _Saver:
    def restore(self, session, save_path):
        from demon_amd import weights as W
        if os.path.exists(save_path + ".index"):
            from demon_amd.tf_checkpoint import load_tf_checkpoint, read_index
            # original model (example.py) or v2 model (example_v2.py --checkpoint): told apart by the dense5 layer
            version = 2 if "netFlow1/dense5/kernel" in read_index(save_path)[0] else 1
            w = load_tf_checkpoint(save_path, list(W.variable_shapes(version=version)))
        elif os.path.exists(save_path + ".npz"):
            w = W.load_npz(save_path + ".npz")
        elif os.environ.get("DEMON_SYNTHETIC_WEIGHTS") in ("1", "2"):
            w = W.synthetic_weights(seed=1, version=int(os.environ["DEMON_SYNTHETIC_WEIGHTS"]))
        else:
            raise IOError("checkpoint %s(.index|.npz) not found (set DEMON_SYNTHETIC_WEIGHTS=1 / =2 for random weights of the original / v2 model)" % save_path)
        # the weights become this session's variables: networks constructed on it before restore() pick them up here, nets of
        # other sessions keep theirs; the first restored set also becomes the process default
        from demon_amd import runtime
        runtime.set_session_weights(session, w)


class _Train:
    Saver = _Saver


train = _Train()
