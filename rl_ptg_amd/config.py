"""Environment configuration: the constants of the reference's config/config_env.yaml and the values its
EnvConfiguration derives from them (src/rl_config_env.py:12-49).

`EnvConfig()` gives the reference's defaults; `EnvConfig.from_yaml(path)` reads a reference-style
config_env.yaml (same keys), `EnvConfig(scenario=1, operation="OP1", ...)` overrides single keys.
"""
import copy

DEFAULTS = {
    "scenario": 2, "operation": "OP2", "price_ahead": 13, "time_step_op": 2, "noise": 10, "eps_len_d": 37,
    "state_change_penalty": 0.0, "sim_step": 600, "raw_modified": "mod",
    "ptg_state_space": {"standby": 0, "cooldown": 1, "startup": 2, "partial_load": 3, "full_load": 4},
    # steady-state operation per load level [off, partial_load, full_load] (config_env.yaml:81-101)
    "meth_stats_load": {
        "OP1": {"Meth_State": [2, 5, 5], "Meth_Action": [6, 10, 11], "Meth_Hot_Cold": [0, 1, 1],
                "Meth_T_cat": [11.0, 451.0, 451.0], "Meth_H2_flow": [0.0, 0.00701, 0.0198],
                "Meth_CH4_flow": [0.0, 0.00172, 0.0048], "Meth_H2_res_flow": [0.0, 0.000054, 0.000151],
                "Meth_H2O_flow": [0.0, 0.0624, 0.458545], "Meth_el_heating": [0.0, 231.0, 350.0]},
        "OP2": {"Meth_State": [2, 5, 5], "Meth_Action": [6, 10, 11], "Meth_Hot_Cold": [0, 1, 1],
                "Meth_T_cat": [11.0, 451.0, 451.0], "Meth_H2_flow": [0.0, 0.0198, 0.0485],
                "Meth_CH4_flow": [0.0, 0.0048, 0.0114], "Meth_H2_res_flow": [0.0, 0.000151, 0.0017],
                "Meth_H2O_flow": [0.0, 0.458545, 1.22], "Meth_el_heating": [0.0, 350.0, 380.0]},
    },
    "ch4_price_fix": 15.0, "heat_price": 4.6, "o2_price": 10.2, "water_price": 6.4, "eeg_el_price": 17.84,
    "H_u_CH4": 35.883, "H_u_H2": 10.783, "h_H2O_evap": 2257, "dt_water": 90, "cp_water": 4.18, "rho_water": 998,
    "convert_mol_to_Nm3": 0.02241407, "Molar_mass_CO2": 44.01, "Molar_mass_H2O": 18.02,
    "min_load_electrolyzer": 0.032, "eta_CHP": 0.38,
    "r_0_values": {"el_price": [0], "gas_price": [10], "eua_price": [50]},
    "t_cat_standby": 188.2, "t_cat_startup_cold": 160, "t_cat_startup_hot": 350,
    "time1_start_p_f": 1201, "time2_start_f_p": 151, "time_p_f": 210, "time_f_p": 126, "time1_p_f_p": 51,
    "time2_p_f_p": 151, "time23_p_f_p": 225, "time3_p_f_p": 301, "time34_p_f_p": 376, "time4_p_f_p": 451,
    "time45_p_f_p": 563, "time5_p_f_p": 675, "time1_f_p_f": 51, "time2_f_p_f": 151, "time23_f_p_f": 225,
    "time3_f_p_f": 301, "time34_f_p_f": 376, "time4_f_p_f": 451, "time45_f_p_f": 526, "time5_f_p_f": 601,
    "i_fully_developed": 12000, "j_fully_developed": 100,
    "el_l_b": -10, "el_u_b": 90, "gas_l_b": 0.4, "gas_u_b": 32, "eua_l_b": 23, "eua_u_b": 98, "T_l_b": 10,
    "T_u_b": 600, "h2_l_b": 0, "ch4_l_b": 0, "h2_res_l_b": 0, "h2o_l_b": 0, "heat_l_b": 0, "heat_u_b": 1800,
}

STATS_NAMES = ['steps_stats', 'el_price_stats', 'gas_price_stats', 'eua_price_stats', 'Meth_State_stats',
               'Meth_Action_stats', 'Meth_Hot_Cold_stats', 'Meth_T_cat_stats', 'Meth_H2_flow_stats',
               'Meth_CH4_flow_stats', 'Meth_H2O_flow_stats', 'Meth_el_heating_stats', 'Meth_ch4_revenues_stats',
               'Meth_steam_revenues_stats', 'Meth_o2_revenues_stats', 'Meth_eua_revenues_stats',
               'Meth_chp_revenues_stats', 'Meth_elec_costs_heating_stats', 'Meth_elec_costs_electrolyzer_stats',
               'Meth_water_costs_stats', 'Meth_reward_stats', 'Meth_cum_reward_stats', 'pot_reward_stats',
               'part_full_stats']

# keys copied verbatim into the env kwargs (src/rl_utils.py:348-360)
KWARG_KEYS = ["noise", "eps_len_d", "sim_step", "time_step_op", "price_ahead", "scenario",
              "convert_mol_to_Nm3", "H_u_CH4", "H_u_H2", "dt_water", "cp_water", "rho_water",
              "Molar_mass_CO2", "Molar_mass_H2O", "h_H2O_evap", "eeg_el_price", "heat_price",
              "o2_price", "water_price", "min_load_electrolyzer", "max_h2_volumeflow", "eta_CHP",
              "t_cat_standby", "t_cat_startup_cold", "t_cat_startup_hot", "time1_start_p_f",
              "time2_start_f_p", "time_p_f", "time_f_p", "time1_p_f_p", "time2_p_f_p",
              "time23_p_f_p", "time3_p_f_p", "time34_p_f_p", "time4_p_f_p", "time45_p_f_p",
              "time5_p_f_p", "time1_f_p_f", "time2_f_p_f", "time23_f_p_f", "time3_f_p_f",
              "time34_f_p_f", "time4_f_p_f", "time45_f_p_f", "time5_f_p_f", "i_fully_developed",
              "j_fully_developed", "el_l_b", "el_u_b", "gas_l_b", "gas_u_b", "eua_l_b", "eua_u_b",
              "T_l_b", "T_u_b", "h2_l_b", "h2_u_b", "ch4_l_b", "ch4_u_b", "h2_res_l_b", "h2_res_u_b",
              "h2o_l_b", "h2o_u_b", "heat_l_b", "heat_u_b", "raw_modified"]


class EnvConfig:
    """Attribute bag like the reference's EnvConfiguration (src/rl_config_env.py:12-49), without file paths."""

    def __init__(self, **overrides):
        d = copy.deepcopy(DEFAULTS)
        unknown = set(overrides) - set(d)
        if unknown:
            raise KeyError(f"unknown config keys: {sorted(unknown)}")
        d.update(overrides)
        self.__dict__.update(d)
        assert self.scenario in [1, 2, 3], f"Specified business scenario ({self.scenario}) must match one of [1,2,3]!"
        assert self.raw_modified in ["raw", "mod"], f"Invalid state design {self.raw_modified}"
        assert self.operation in ["OP1", "OP2"], f"Invalid load level {self.operation}"
        self.train_len_d = None
        self.meth_stats_load = self.meth_stats_load[self.operation]
        self.max_h2_volumeflow = self.convert_mol_to_Nm3 * self.meth_stats_load["Meth_H2_flow"][2]
        self.h2_u_b = self.meth_stats_load["Meth_H2_flow"][2]
        self.ch4_u_b = self.meth_stats_load["Meth_CH4_flow"][2]
        self.h2_res_u_b = self.meth_stats_load["Meth_H2_res_flow"][2]
        self.h2o_u_b = self.meth_stats_load["Meth_H2O_flow"][2]
        self.stats_names = list(STATS_NAMES)

    @classmethod
    def from_yaml(cls, path, **overrides):
        import yaml
        with open(path) as f:
            y = yaml.safe_load(f)
        keep = {k: v for k, v in y.items() if k in DEFAULTS}
        keep.update(overrides)
        return cls(**keep)
