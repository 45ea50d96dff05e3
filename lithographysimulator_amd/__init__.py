"""MI355X-native Abbe aerial-image engine behind the object API of
quarterwave0/LithographySimulator (Mask / LightSource / Pupil / abbeImage)."""
from .imageformation import (PlanCache, abbeImage, abbeIntensity, bossungCurves, calculateFFTAerial,   # noqa: F401
                             embeddedSize, postProcess, resistContour)
from ._native import engineOptions                                                      # noqa: F401
from .layout import (GdsLibrary, flattenLayout, maskFromGDSII, rasterizeLayout, readGDSII,  # noqa: F401
                     writeGDSII)
from .lightsource import LightSource, sourceShifts, sourceShiftsAsync                                      # noqa: F401
from .mask import Mask                                                                  # noqa: F401
from .pupil import (OSAindexToMN, Pupil, generatePhi, generateWavefrontError,           # noqa: F401
                    generateZ, throughFocusPupils)

__all__ = ["Mask", "LightSource", "Pupil", "abbeImage", "abbeIntensity", "calculateFFTAerial", "postProcess", "resistContour", "bossungCurves", "PlanCache", "engineOptions", "embeddedSize",
           "sourceShifts", "sourceShiftsAsync", "OSAindexToMN", "generateWavefrontError", "generatePhi", "generateZ",
           "throughFocusPupils", "readGDSII", "writeGDSII", "flattenLayout", "rasterizeLayout", "maskFromGDSII", "GdsLibrary"]
