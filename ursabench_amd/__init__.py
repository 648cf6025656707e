"""ursabench_amd — MI355X-native SG-MCMC + Bayesian-model-averaging engine behind URSABench's
sampler (`inference`) and task (`tasks`) plug-in surfaces. See DESIGN.md / INTEGRATION.md.

    from ursabench_amd import inference, tasks, util, models
"""
__version__ = '0.1.0'
