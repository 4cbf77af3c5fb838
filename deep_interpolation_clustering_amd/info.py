"""Surface constants of the reference pipeline (info.py:1-41): paths, cohort names, the six vitals
and their physiologic ranges, and the metric names the trainers checkpoint on."""
import os

BASE_PATH = os.path.dirname(os.getcwd())        # data lives in ../Data relative to the run directory (info.py:2)
USE_FEATURES = ['sbp', 'dbp', 'heartRate', 'temperature', 'spo2', 'respiratory']
COHORTS = ['training', 'validation', 'testing']
COHORT2SCOPE = dict(zip(COHORTS, ('train', 'valid', 'test')))
DATA_DICT_KEYS = ['feat', 'time_step', 'padding_mask', 'encounter_id']
MIN_MAX_VALUES = {
    'sbp': [20, 300], 'dbp': [5, 225], 'heartRate': [0, 300],
    'temperature': [24, 45], 'spo2': [0, 100], 'respiratory': [0, 60],
}
LEGEND_INFO = {str(i): 'Phenotype ' + chr(ord('A') + i) for i in range(10)}
PALETTE_INFO = {0: '#9b59b6', 1: '#3498db', 2: '#8de5a1', 3: '#e74c3c', 4: '#34495e', 5: '#2ecc71'}

METRICS = ['loss', 'ae_mse', 'delta']            # a checkpoint directory per metric (utils.create_weight_dir)
MIN_METRICS = ['loss', 'ae_mse', 'delta']        # lower is better
MAX_METRICS = []
SUMMARY_ITEMS = ['lr', 'kl', 'fake_detection']
