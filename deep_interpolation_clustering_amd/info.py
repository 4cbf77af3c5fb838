"""Surface constants of the reference pipeline (upstream info.py): where the data lives, the cohort names, the monitored
vitals with their physiologic ranges (used to map reconstructions back to physical units) and the metric names the trainers
keep a best checkpoint for.  Everything is derived from the two small tables below."""
import os

# the run directory is <base>/<something>; pickles are read from <base>/Data/model_data/split_processed
BASE_PATH = os.path.dirname(os.getcwd())

# (name in the cohort tables, lowest / highest physiologic value)
_VITALS = (
    ('sbp', 20, 300),
    ('dbp', 5, 225),
    ('heartRate', 0, 300),
    ('temperature', 24, 45),
    ('spo2', 0, 100),
    ('respiratory', 0, 60),
)
USE_FEATURES = [name for name, _, _ in _VITALS]
MIN_MAX_VALUES = {name: [lo, hi] for name, lo, hi in _VITALS}

# (pickle / dump file stem, short scope name used in logs and summaries)
_SPLITS = (('training', 'train'), ('validation', 'valid'), ('testing', 'test'))
COHORTS = [stem for stem, _ in _SPLITS]
COHORT2SCOPE = dict(_SPLITS)
DATA_DICT_KEYS = ['feat', 'time_step', 'padding_mask', 'encounter_id']

# presentation (p4 tables): cluster id -> display name / colour
LEGEND_INFO = {str(i): 'Phenotype ' + chr(ord('A') + i) for i in range(10)}
PALETTE_INFO = dict(enumerate(('#9b59b6', '#3498db', '#8de5a1', '#e74c3c', '#34495e', '#2ecc71')))

# one best-checkpoint directory per metric; all three are minimised
METRICS = ['loss', 'ae_mse', 'delta']
MIN_METRICS = list(METRICS)
MAX_METRICS = []
SUMMARY_ITEMS = ['lr', 'kl', 'fake_detection']
