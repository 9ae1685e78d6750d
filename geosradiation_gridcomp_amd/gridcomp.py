"""Argument orders of the GridComp data-path entry points (include/geosrad.h, the GEOSRAD_LWD_* / SWD_* / LWU_* / SWU_* / RT_*
enums) and the constants MAPL supplies to the reference (MAPL is not part of the reference repository; these are the values of
MAPL_Constants and are only defaults for callers that do not pass their own).

Reference: GEOSirrad_GridComp/GEOS_IrradGridComp.F90 (LW_Driver :3188-3615, Update_Flx :3796-3999),
GEOSsolar_GridComp/GEOS_SolarGridComp.F90 (SORADCORE :6113-6450, UPDATE_EXPORT :7540-7579), GEOS_RadiationGridComp.F90:798-819.
"""

LWD_IN = ["PLE", "PL", "T", "Q", "O3", "CH4", "N2O", "CO2_3D", "CFC11", "CFC12", "HCFC22", "FCLD", "CWC_LIQ", "CWC_ICE", "REFF_LIQ",
          "REFF_ICE", "TAUA", "SSAA", "TS", "EMIS", "LATS", "T2M"]
LWD_CONST = ["CO2_FIXED", "O2", "CCL4", "AIRMW", "H2OMW", "O3MW", "RGAS", "GRAV"]
LWD_OUT = ["FLXU_INT", "FLXD_INT", "FLCU_INT", "FLCD_INT", "DFDTS", "DFDTSC", "DFDTSNA", "DFDTSCNA", "FLX_INT", "FLC_INT", "SFCEM_INT",
           "TS_INT", "CLDTTLW", "CLDHILW", "CLDMDLW", "CLDLOLW", "OLRB", "DOLRB"]
LWD_OUT_3D = LWD_OUT[:10]

# RATS diagnostics (IRR:3389-3469): gas codes = include/geosrad.h GEOSRAD_RAT_*; the solver-side array each one removes
RAT_GAS = ["H2O", "O3", "CO2", "CH4", "N2O", "CFC11", "CFC12", "HCFC22"]
RAT_VMR = dict(H2O="h2ovmr", O3="o3vmr", CO2="co2vmr", CH4="ch4vmr", N2O="n2ovmr", CFC11="cfc11vmr", CFC12="cfc12vmr", HCFC22="cfc22vmr")
LWD_RAT_OUT = ["FLXU_RAT", "FLXD_RAT", "FLX_RAT", "DFDTS_RAT", "SFCEM_RAT"]
# RATS exports of Update_Flx (IRR:4036-4120), GEOSRAD_LWR_*: per listed gas "dOLR_<gas>" ...; arrays with the gas slowest
LWR_IN = ["FLX_INT", "SFCEM_INT", "DFDTS", "FLX_RAT", "SFCEM_RAT", "DFDTS_RAT"]
LWR_OUT = ["dOLR", "dLWS", "dFLNS", "dSFCEM", "NETTRAP", "COLTRAP", "FLX", "DFDTS_OUT"]

LWC_IN = ["FLXU_INT", "FLCU_INT", "FLAU_INT", "FLXAU_INT", "FLXD_INT", "FLCD_INT", "FLAD_INT", "FLXAD_INT", "DFDTS", "TS"]
LWC_OUT = ["SFCEM_INT", "FLX_INT", "FLXA_INT", "FLC_INT", "FLA_INT", "DFDTSC", "DFDTSNA", "DFDTSCNA", "TS_INT"]

SWD_IN = ["PLE", "PL", "T", "Q", "O3", "CH4", "CL", "TS", "QQ_ICE", "QQ_LIQ", "RR_ICE", "RR_LIQ", "TAUA", "SSAA", "ASYA", "ZT", "ALAT",
          "ALBVR", "ALBVF", "ALBNR", "ALBNF"]
SWD_CONST = ["CO2", "O2", "AIRMW", "H2OMW", "O3MW", "RGAS", "GRAV", "UNDEF"]
SWD_OUT = ["FSW", "FSC", "FSWU", "FSCU", "NIRR", "NIRF", "PARR", "PARF", "UVRR", "UVRF", "FSWBAND", "CLDTS", "CLDHS", "CLDMS", "CLDLS",
           "COTTP", "COTHP", "COTMP", "COTLP", "FSWNA", "FSCNA", "FSWUNA", "FSCUNA", "FSWBANDNA"]

# Chou-Suarez branch of SORADCORE (SOL:4484-4572, SHRTWAVE :6597-6672): GEOSRAD_SWC_*
SWC_IN = ["PLE", "T", "Q", "OX", "CL", "QI", "QL", "QR", "QS", "RI", "RL", "RR", "RS", "TAUA", "SSAA", "ASYA", "ZT", "ALBVR", "ALBVF",
          "ALBNR", "ALBNF"]
SWC_CONST = ["CO2", "O3MW", "AIRMW", "UNDEF"]
SWC_OUT = ["FSW", "FSC", "FSWU", "FSCU", "NIRR", "NIRF", "PARR", "PARF", "UVRR", "UVRF", "FSWBAND", "DRBAND", "DFBAND"]

LWU_IN = ["TSINST", "TS_INT", "SFCEM_INT", "FCLD", "FLX_INT", "FLXA_INT", "FLC_INT", "FLA_INT", "FLXU_INT", "FLXAU_INT", "FLCU_INT",
          "FLAU_INT", "FLXD_INT", "FLXAD_INT", "FLCD_INT", "FLAD_INT", "DFDTS", "DFDTSNA", "DFDTSC", "DFDTSCNA"]
LWU_IN_NA = ["FLXA_INT", "FLA_INT", "FLXAU_INT", "FLAU_INT", "FLXAD_INT", "FLAD_INT", "DFDTSNA", "DFDTSCNA"]
LWU_OUT_3D = ["FLX", "FLXA", "FLC", "FLA", "FLXU", "FLXAU", "FLCU", "FLAU", "FLXD", "FLXAD", "FLCD", "FLAD"]
LWU_OUT_2D = ["OLR", "OLRA", "OLC", "OLA", "OLCC5", "DSFDTS", "SFCEM", "LWS", "LWSA", "LCS", "LAS", "LCSC5", "FLNS", "FLNSNA", "FLNSC",
              "FLNSA", "DSFDTS0", "SFCEM0", "TSREFF", "CLDTT"]
LWU_OUT = LWU_OUT_3D + LWU_OUT_2D

SWU_IN = ["SLR", "FSWN", "FSCN", "FSWNAN", "FSCNAN", "FSWUN", "FSCUN", "FSWUNAN", "FSCUNAN", "FSWBANDN", "FSWBANDNAN"]
SWU_OUT_3D = ["FSW", "FSC", "FSWNA", "FSCNA", "FSWU", "FSCU", "FSWUNA", "FSCUNA", "FSWD", "FSCD", "FSWDNA", "FSCDNA"]
SWU_OUT_BAND = ["FSWBAND", "FSWBANDNA"]
SWU_OUT_2D = ["RSR", "RSC", "RSRNA", "RSCNA", "RSRS", "RSCS", "RSRSNA", "RSCSNA", "OSR", "OSRCLR", "OSRNA", "OSRCNA"]
SWU_OUT = SWU_OUT_3D + SWU_OUT_BAND + SWU_OUT_2D

RT_IN = ["PLE", "FLW", "FSW", "FLWCLR", "FSWCLR", "FSWNA", "FLA", "FSCNA", "DSFDTS", "SFCEM", "TRD"]
RT_OUT_3D = ["DTDT", "RADLW", "RADSW", "RADLWC", "RADSWC", "RADSWNA", "RADLWCNA", "RADSWCNA"]
RT_OUT_2D = ["BLW", "ALW", "RADSRF"]
RT_OUT = RT_OUT_3D + RT_OUT_2D

# MAPL_Constants (not in the reference repository): defaults only
MAPL = {"AIRMW": 28.965, "H2OMW": 18.015, "O3MW": 47.9982, "RUNIV": 8314.47, "GRAV": 9.80665, "CP": 1004.6830, "UNDEF": 1.0e15}
MAPL["RGAS"] = MAPL["RUNIV"] / MAPL["AIRMW"]
# trace-gas scalars of the Irrad / Solar resource files (IRR:1594-1597)
GAS = {"CO2_FIXED": 4.0e-4, "CO2": 4.0e-4, "O2": 0.2090029, "CCL4": 0.1105000e-09}


def swc_consts(co2=None, **over):
    d = dict(CO2=GAS["CO2"] if co2 is None else co2, O3MW=MAPL["O3MW"], AIRMW=MAPL["AIRMW"], UNDEF=MAPL["UNDEF"])
    d.update(over)
    return [d[k] for k in SWC_CONST]


def lwd_consts(**over):
    d = dict(CO2_FIXED=GAS["CO2_FIXED"], O2=GAS["O2"], CCL4=GAS["CCL4"], AIRMW=MAPL["AIRMW"], H2OMW=MAPL["H2OMW"], O3MW=MAPL["O3MW"],
             RGAS=MAPL["RGAS"], GRAV=MAPL["GRAV"])
    d.update(over)
    return [float(d[k]) for k in LWD_CONST]


def swd_consts(**over):
    d = dict(CO2=GAS["CO2"], O2=GAS["O2"], AIRMW=MAPL["AIRMW"], H2OMW=MAPL["H2OMW"], O3MW=MAPL["O3MW"], RGAS=MAPL["RGAS"],
             GRAV=MAPL["GRAV"], UNDEF=MAPL["UNDEF"])
    d.update(over)
    return [float(d[k]) for k in SWD_CONST]

# rrlw_wvn: band limits of RRTMG_LW [cm-1] (LW/modules/rrlw_wvn.F90, set in rrtmg_lw_init.F90)
LW_WAVENUM1 = [10., 350., 500., 630., 700., 820., 980., 1080., 1180., 1390., 1480., 1800., 2080., 2250., 2380., 2600.]
LW_WAVENUM2 = [350., 500., 630., 700., 820., 980., 1080., 1180., 1390., 1480., 1800., 2080., 2250., 2380., 2600., 3250.]

# 2-D block of UPDATE_EXPORT (SOL:7403-7533), GEOSRAD_SWS_*
SWS_IN = ["SLR", "ZTH", "ALBVF", "ALBVR", "ALBNF", "ALBNR", "DRUVRN", "DFUVRN", "DRPARN", "DFPARN", "DRNIRN", "DFNIRN", "FSWN", "FSCN", "FSWNAN",
          "FSCNAN"]
SWS_OUT = ["ALBVF_X", "ALBVR_X", "ALBNF_X", "ALBNR_X", "ALBEDO", "SLRTP", "DRUVR", "DFUVR", "DRPAR", "DFPAR", "DRNIR", "DFNIR", "DRNUVR", "DRNPAR",
           "DRNNIR", "SLRSF", "SLRSFC", "SLRSFNA", "SLRSFCNA", "SLRSUF", "SLRSUFC", "SLRSUFNA", "SLRSUFCNA"]
