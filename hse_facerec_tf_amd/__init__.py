"""hse_facerec_tf_amd -- MI355X-native engine for the feature-extract hot path of
av-savchenko/HSE_FaceRec_tf (facerec_test.py TensorFlowInference / facial_analysis.py
age_gender_fun), behind the reference's own Python API.

Importing the package is cheap and GPU-free; constructing an extractor needs libhsefr.so and
a gfx950 device, and fails loudly otherwise (there is no CPU path in this package).
"""
from .graphdef import read_graph  # noqa: F401
from .lowering import lower_graph, LoweringError  # noqa: F401
from .tf_inference import (TensorFlowInference, extract_dataset, extract_gallery_probe, get_files,  # noqa: F401
                           get_tf_face_recognizer, load_graph)
from .facial_analysis import FacialImageProcessing  # noqa: F401

__version__ = "0.1.0"
