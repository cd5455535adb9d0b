"""Configuration surface of the two-stream path: the names and values of the reference's
``Sheet03/parameters.py:2-46``, kept as module constants (the reference has no CLI and no
environment variables).  Site-specific absolute paths of the reference (``:26-33``) become
overridable through ``VA_DATA_ROOT``; everything else is value-identical.
"""
import os

_ROOT = os.environ.get("VA_DATA_ROOT", "./data")

# Some parameters (Sheet03/parameters.py:2-20)
VIDEO_FRAME_SAMPLE_RATE = 10
CONVERT = False
VIDEO_INPUT_FRAME_COUNT = 3
VIDEO_INPUT_FLOW_COUNT = 10
SPATIAL_BATCH_SIZE = 60
TEMPORAL_BATCH_SIZE = 32
NWORKERS_LOADER = 4
SHUFFLE_LOADER = True
CROP_SIZE_TF = 224
HORIZONTAL_FLIP_TF = True
NORM_MEANS_TF = [0.485, 0.456, 0.406]
NORM_STDS_TF = [0.229, 0.224, 0.225]
NACTION_CLASSES = 101
NEPOCHS = 25
INITIAL_LR = 0.1
MOMENTUM_VAL = 0.9
MILESTONES_LR = [10, 20]
VIDEO_DESCRIPTOR_DIM = 256
N_FIXED_LAYERS = 5
COLOR_JITTERS = [0, 0, 0, 0]  # brightness, contrast, saturation, hue

# Some constants (Sheet03/parameters.py:23-46)
VIDEO_EXTN = ".avi"
FRAME_EXTN = ".jpg"
DATA_DIR = os.path.join(_ROOT, "mini-UCF-101")
FLOW_DATA_DIR = os.path.join(_ROOT, "mini-ucf101_flow_img_tvl1_gpu")
FRAMES_DIR_TRAIN = os.path.join(_ROOT, "mini-UCF-101-frames-train")
FRAMES_DIR_TEST = os.path.join(_ROOT, "mini-UCF-101-frames-test")
VIDEOLIST_TRAIN = os.path.join(_ROOT, "demoTrain.txt")
VIDEOLIST_TEST = os.path.join(_ROOT, "demoTest.txt")
ACTIONLABEL_FILE = os.path.join(_ROOT, "classInd.txt")
CHECKPOINT_DIR = os.path.join(_ROOT, "checkpoints/")
SPATIAL_CKP_FILE = "spatial_ckp.pth.tar"
SPATIAL_BEST_FILE = "spatial_best.pth.tar"
MOTION_CKP_FILE = "temporal_ckp.pth.tar"
MOTION_BEST_FILE = "temporal_best.pth.tar"
X_PREFIX_FLOW = "flow_x_"
Y_PREFIX_FLOW = "flow_y_"
TEMPORAL_TRAIN_CSV_LOC = "./temporal_train.csv"
TEMPORAL_TEST_CSV_LOC = "./temporal_test.csv"
SPATIAL_TEST_CSV_LOC = "./spatial_test.csv"
SPATIAL_TRAIN_CSV_LOC = "./spatial_train.csv"
SPATIAL_PERFORMANCE_LOC = "./spatial_performance.csv"
TEMPORAL_PERFORMANCE_LOC = "./temporal_performance.csv"
SVM_FILE = "svm_classifier.pkl"
