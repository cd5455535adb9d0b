"""Configuration surface of the two-stream path.

The reference configures everything through the module constants of ``Sheet03/parameters.py`` (no
CLI, no environment variables) and star-imports them everywhere; this module keeps every one of
those NAMES and VALUES (``Sheet03/parameters.py:2-21`` hyper-parameters, ``:23-46`` file layout) so
that code written against the reference's config keeps working.  Only the site-specific absolute
paths of ``:26-33`` differ: they hang off ``VA_DATA_ROOT`` (default ``./data``).
"""
import os as _os


def _under_root(*parts):
    return _os.path.join(_os.environ.get("VA_DATA_ROOT", "./data"), *parts)


# --- sampling of frames / flow fields per video ------------------------------ (:2-5)
VIDEO_FRAME_SAMPLE_RATE, CONVERT = 10, False
VIDEO_INPUT_FRAME_COUNT, VIDEO_INPUT_FLOW_COUNT = 3, 10

# --- data loading ------------------------------------------------------------ (:6-9)
SPATIAL_BATCH_SIZE, TEMPORAL_BATCH_SIZE = 60, 32
NWORKERS_LOADER, SHUFFLE_LOADER = 4, True

# --- image transforms -------------------------------------------------------- (:10-13, :21)
CROP_SIZE_TF, HORIZONTAL_FLIP_TF = 224, True
NORM_MEANS_TF = [0.485, 0.456, 0.406]
NORM_STDS_TF = [0.229, 0.224, 0.225]
COLOR_JITTERS = [0, 0, 0, 0]  # brightness, contrast, saturation, hue: the identity

# --- model / optimisation ---------------------------------------------------- (:14-20)
NACTION_CLASSES, VIDEO_DESCRIPTOR_DIM, N_FIXED_LAYERS = 101, 256, 5
NEPOCHS, INITIAL_LR, MOMENTUM_VAL, MILESTONES_LR = 25, 0.1, 0.9, [10, 20]

# --- file name conventions --------------------------------------------------- (:24-25, :38-39)
VIDEO_EXTN, FRAME_EXTN = ".avi", ".jpg"
X_PREFIX_FLOW, Y_PREFIX_FLOW = "flow_x_", "flow_y_"

# --- data locations (site specific in the reference, :26-33) ------------------------------------
DATA_DIR = _under_root("mini-UCF-101")
FLOW_DATA_DIR = _under_root("mini-ucf101_flow_img_tvl1_gpu")
FRAMES_DIR_TRAIN, FRAMES_DIR_TEST = _under_root("mini-UCF-101-frames-train"), _under_root("mini-UCF-101-frames-test")
VIDEOLIST_TRAIN, VIDEOLIST_TEST = _under_root("demoTrain.txt"), _under_root("demoTest.txt")
ACTIONLABEL_FILE = _under_root("classInd.txt")
CHECKPOINT_DIR = _under_root("checkpoints") + "/"

# --- outputs ----------------------------------------------------------------- (:34-37, :40-46)
SPATIAL_CKP_FILE, SPATIAL_BEST_FILE = "spatial_ckp.pth.tar", "spatial_best.pth.tar"
MOTION_CKP_FILE, MOTION_BEST_FILE = "temporal_ckp.pth.tar", "temporal_best.pth.tar"
TEMPORAL_TRAIN_CSV_LOC, TEMPORAL_TEST_CSV_LOC = "./temporal_train.csv", "./temporal_test.csv"
SPATIAL_TRAIN_CSV_LOC, SPATIAL_TEST_CSV_LOC = "./spatial_train.csv", "./spatial_test.csv"
SPATIAL_PERFORMANCE_LOC, TEMPORAL_PERFORMANCE_LOC = "./spatial_performance.csv", "./temporal_performance.csv"
SVM_FILE = "svm_classifier.pkl"
