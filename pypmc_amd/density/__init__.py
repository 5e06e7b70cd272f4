from . import base, gauss, student_t, mixture
