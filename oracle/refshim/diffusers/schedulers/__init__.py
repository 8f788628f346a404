from .. import EulerDiscreteScheduler
