"""Process-wide settings: the knobs the reference keeps as class attributes of ``Gs`` / string tags of ``Gsv``
(reference ``Helpers/GlobalSettings.py:6-109``).  Only the attributes the hypergraph path and its driver read
are carried; values are the reference defaults.
"""
import torch.nn as nn


class Gsv:
    mean, activation, rnn = 'mean', 'activation', 'rnn'
    concat, product = 'concatenation', 'product'
    graph_uqi, graph_only_uq, graph_only_ui, graph_only_qi = 'uqi', 'uq', 'ui', 'qi'


class Gs:
    use_valid_dataset = True
    adjust_learning_rate = True          # lr *= 0.98 per epoch once avg loss < 0.008 while lr > 4e-4
    lambda_muq_for_hem = 0.5

    batch_size = 100                     # positives per batch; rows per batch = batch_size * (1 + negatives)
    batch_size_times = 1
    learning_rate = 0.001
    embedding_size = 32
    weight_decay = 0

    graph_completeness = Gsv.graph_uqi
    long_tail_stat_fn = None

    random_negative_sample_size = 10
    non_random_negative_sample_size = 0
    negative_sample_size = random_negative_sample_size + non_random_negative_sample_size

    class Query:
        transform = Gsv.mean             # the only transform on the path (EmbeddingLayers.py:37-38)
        transform_activation = nn.ReLU

    class Prediction:
        use_cosine_similarity = False

    class Dataset:
        user_history_limit = 500

    class Debug:
        show_highorder_embedding_info = False
        _calculate_embedding_info = False
        _calculate_highorder_info = False
