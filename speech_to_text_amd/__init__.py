"""MI355X-native MFCC engine: a drop-in for the librosa MFCC path of jvargeaux/speech_to_text
(/root/reference/src/preprocess.py:131-167).  The product path is libmfcc_hip.so (HIP, gfx950);
there is no CPU fallback."""
from . import cmvn, hdf5_min, shard
from .config import MfccConfig, ParameterError, make_config
from .feature import MfccPlan, get_plan, melspectrogram, mfcc, mfcc_batch, spectrogram
from .preprocess import Preprocessor, read_feature_file, write_feature_file

__all__ = ["MfccConfig", "ParameterError", "make_config", "MfccPlan", "get_plan", "mfcc",
           "mfcc_batch", "melspectrogram", "spectrogram", "Preprocessor", "read_feature_file",
           "write_feature_file", "cmvn", "shard", "hdf5_min"]
