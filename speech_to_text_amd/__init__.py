"""MI355X-native MFCC engine: a drop-in for the librosa MFCC path of jvargeaux/speech_to_text
(/root/reference/src/preprocess.py:131-167).  The product path is libmfcc_hip.so (HIP, gfx950);
there is no CPU fallback."""
from . import cmvn, hdf5_min, shard, sink
from .config import MfccConfig, ParameterError, make_config
from .feature import (MfccPlan, amplitude_to_db, delta, delta_batch, get_plan, melspectrogram, mfcc, mfcc_batch, power_to_db,
                      spectrogram)
from .preprocess import Preprocessor, read_feature_file, write_feature_file
from .sink import FeatureDataset, PackedReader, load_training_items, write_packed

__all__ = ["MfccConfig", "ParameterError", "make_config", "MfccPlan", "get_plan", "mfcc",
           "mfcc_batch", "melspectrogram", "spectrogram", "power_to_db", "amplitude_to_db", "delta", "delta_batch", "Preprocessor", "read_feature_file",
           "write_feature_file", "FeatureDataset", "PackedReader", "load_training_items", "write_packed",
           "cmvn", "shard", "hdf5_min", "sink"]
